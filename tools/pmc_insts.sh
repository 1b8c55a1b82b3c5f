#!/bin/bash
# Instruction-mix PMC pass for one or more library variants: tools/pmc_insts.sh "<bench args>" libA.so libB.so ...
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  OUT=$R/gpurun_out/pmc_insts/$lib
  mkdir -p $OUT
  AACGPU_LIB=$R/aac.js_amd/csrc/variants/$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d $OUT -- python3 $R/bench.py --steps 40 --warmup 10 --precondition-ms 0 --no-cpu-baseline $ARGS > $OUT/log.txt 2>&1
  python3 - $OUT $lib <<'PY'
import sys, glob, csv, collections
out, lib = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "imdct_run" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / max(1, len(acc["SQ_WAVES"]))
print(lib, " ".join("%s/wave %.1f" % (k.replace("SQ_INSTS_", ""), (sum(v) / len(v)) / w) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
done
