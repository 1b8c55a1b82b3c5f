#!/bin/bash
# Quick PMC passes (instruction mix, waits, LDS) of the run kernel bench.py launches: tools/pmc_quick.sh <tag> [bench args...]
TAG=${1:-q}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --precondition-ms 0 --no-cpu-baseline --no-parity "$@" > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "imdct_run" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    w = sum(d["SQ_WAVES"]) / max(1, len(d["SQ_WAVES"]))
    print("==", k, "waves/launch %.0f" % w)
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print("  %-26s %14.0f  per wave %10.1f" % (c, m, m / w))
PY
