#!/bin/bash
# Interleaved A/B of library builds on ONE GPU box: tools/ab.sh "<bench args>" libA.so libB.so ...
# (variants live in aac.js_amd/csrc/variants/, built with `make -C aac.js_amd/csrc variant NAME=x`)
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for lib in "$@"; do
    AACGPU_LIB=$R/aac.js_amd/csrc/variants/$lib python3 $R/bench.py --steps ${AB_STEPS:-4000} --warmup 200 --no-cpu-baseline --no-parity $ARGS 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %-28s us/step %.3f (min %.3f)  host %.2f' % ('$lib', '$ARGS', d['ms_per_step']*1e3, d['timing']['ms_per_step_min']*1e3, d['roofline']['host_enqueue_us_per_step']))"
  done
done
