#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py, CSV summaries into gpurun_out/.
# usage: tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2000 --warmup 200 --repeats 5 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
# pipelined launches overlap: the spacing of the trace's rows, not a row's own duration, is what a launch costs
python3 $R/tools/kernel_intervals.py $OUT/trace > $OUT/intervals.txt 2>&1
# PMC passes: separate runs, counters only (no trace domains)
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 40 --warmup 10 --repeats 2 --precondition-ms 0 --no-cpu-baseline --no-parity "$@" > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/prof_summary.py $OUT
