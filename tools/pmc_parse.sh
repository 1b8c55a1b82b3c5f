#!/bin/bash
# Instruction mix of the device front end's kernel per wave: tools/pmc_parse.sh [frames]
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4096}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_parse
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $OUT -- python3 $R/tools/parse_rate.py --frames $N --steps 3 --warmup 1 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "parse_frames" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
print("waves", w, " ".join("%s/wave %.1f" % (k.replace("SQ_INSTS_", ""), (sum(v) / len(v)) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
