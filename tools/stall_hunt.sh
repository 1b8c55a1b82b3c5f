#!/bin/bash
# Runs on the GPU box (via gpurun), FIRST thing in a call: the two tests that once sat in a wait for ten minutes as the first GPU
# work of a fresh box (DESIGN.md "An unexplained stall, twice"), now with every host wait bounded (AACG_ERR_TIMEOUT after 30 s and a
# dump of what was in flight, aacg_last_error) and a two-minute limit per test.  One campaign = one fresh box; the log says what
# the box was and how long each repetition took.  usage: tools/stall_hunt.sh [repetitions, default 3]
set -u
N=${1:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stall_hunt
mkdir -p $OUT
LOG=$OUT/$(date +%Y%m%dT%H%M%S)_$(hostname | tr -c 'A-Za-z0-9' '_').log
cd $R
{
  echo "campaign $(date -u +%FT%TZ) host $(hostname) uptime $(cut -d' ' -f1 /proc/uptime)s kernel $(uname -r)"
  for i in $(seq 1 $N); do
    t0=$(date +%s.%N)
    timeout 400 python -m pytest tests/test_cce_spec.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout 120 \
        -k "test_gpu_coupling_plan_and_errors or test_pipelined_launches_equal_the_serialised_route_bit_for_bit" > $OUT/.last_hunt.txt 2>&1
    rc=$?
    t1=$(date +%s.%N)
    echo "repetition $i: rc $rc in $(python3 -c "print('%.1f' % ($t1 - $t0))") s: $(tail -1 $OUT/.last_hunt.txt)"
    if [ $rc -ne 0 ]; then echo "---- output of the failing repetition"; cat $OUT/.last_hunt.txt; fi
  done
} > $LOG 2>&1
rm -f $OUT/.last_hunt.txt
cat $LOG
