R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_parse2
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT -- python3 $R/tools/parse_rate.py --frames 4096 --steps 3 --warmup 1 --uniform > $OUT/log.txt 2>&1 < /dev/null
python3 - $OUT <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "parse_frames" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
print("waves", w, " ".join("%s/wave %.0f" % (k, (sum(v) / len(v)) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
tail -2 $OUT/log.txt | cut -c1-200
