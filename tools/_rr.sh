node tools/readchunk_rate.js --streams 256 24 2>&1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('shared_engine_resident','shared_engine_resident_ring8'):
    print(k, d[k])
print(d.get('resident_same_checksum'))"
