#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
one() { # label libdir ablate args
  local out
  if [ -n "$2" ]; then out=$(LD_LIBRARY_PATH=$2 AACG_ABLATE=$3 tools/micro/pipe_drive --repeats 2 $4 2>/dev/null | tail -1); else out=$(tools/micro/pipe_drive --repeats 2 $4 2>/dev/null | tail -1); fi
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('  %-70s %6.2f us  %s' % (sys.argv[2], d['us_per_launch_events'], d['kernel']))" "$out" "$1"
}
D1=$(mktemp -d); cp aac.js_amd/csrc/variants/half8slots.so $D1/libaacgpu.so
D2=$(mktemp -d); cp aac.js_amd/csrc/variants/profile.so $D2/libaacgpu.so
for r in 1 2; do
one "16 waves (shipped)" "" 0 ""
one "8 waves, 7 slots (two per CU)" "" 0 "--half"
one "8 waves, 8 slots (one per CU: 88 KB)" $D1 0 "--half"
one "profile, 8 waves" $D2 0 "--half"
one "profile, 8 waves, flat priorities (64)" $D2 64 "--half"
one "profile, 8 waves, two levels (32)" $D2 32 "--half"
one "profile, 8 waves, no load stagger (128)" $D2 128 "--half"
one "profile, 8 waves, dequant skipped (8)" $D2 8 "--half"
one "profile, 8 waves, epilogue skipped (2)" $D2 2 "--half"
one "profile, 16 waves" $D2 0 ""
done
