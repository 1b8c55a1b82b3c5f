#!/bin/bash
# Runs on the GPU box (via gpurun).  What a launch of the headline route would cost if parts of it were free: the profile build's
# work-skipping switches (aacg_kernels.h, AACG_ABLATE; they do not exist in the library that ships) under the C driver
# (tools/micro/pipe_drive), interleaved with the shipped library on the same box.  The PCM of a skipping run is wrong on purpose.
#   8  the dequantisation's arithmetic skipped: the spectra are the quantised integers converted to float (the loads of spectra
#      and band words, the staging stores and everything behind them stay)
#   2  the epilogue skipped: the wait for the predecessor's tail, the overlap-add and the PCM stores
# usage: tools/floor.sh [rounds, default 2]  > profiles/rNN_dequant_floor.txt
set -u
N=${1:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
[ -f aac.js_amd/csrc/variants/profile.so ] || { echo "no profile build (make -C aac.js_amd/csrc profile)"; exit 1; }
D=$(mktemp -d)
cp aac.js_amd/csrc/variants/profile.so $D/libaacgpu.so
one() { # label, library dir or "", AACG_ABLATE, extra args
  local out
  if [ -n "$2" ]; then out=$(LD_LIBRARY_PATH=$2 AACG_ABLATE=$3 tools/micro/pipe_drive --repeats 3 $4 2>/dev/null | tail -1)
  else out=$(tools/micro/pipe_drive --repeats 3 $4 2>/dev/null | tail -1); fi
  python3 -c "
import json,sys
d=json.loads(sys.argv[1])
print('  %-92s %6.2f us per launch   (%s)' % (sys.argv[2], d['us_per_launch_events'] or d['us_per_launch_host_clock'], d['kernel']))" "$out" "$1"
}
echo "tools/floor.sh: config 2 (4096 stereo frames per launch, int16 seam -> f32 PCM), 3 x 30 000 launches per line, $N rounds interleaved, one box"
for r in $(seq 1 $N); do
  echo "round $r, launches overlapped (aacg_decode_pipelined):"
  one "the library that ships" "" 0 ""
  one "profile build, nothing skipped" $D 0 ""
  one "profile build, dequantisation arithmetic skipped (AACG_ABLATE=8)" $D 8 ""
  one "profile build, epilogue skipped: tail wait, overlap-add, PCM stores (AACG_ABLATE=2)" $D 2 ""
  one "profile build, both skipped (AACG_ABLATE=10)" $D 10 ""
  echo "round $r, launch behind launch (aacg_decode_device):"
  one "the library that ships" "" 0 "--serial"
  one "profile build, nothing skipped" $D 0 "--serial"
  one "profile build, dequantisation arithmetic skipped (AACG_ABLATE=8)" $D 8 "--serial"
  one "profile build, epilogue skipped: tail wait, overlap-add, PCM stores (AACG_ABLATE=2)" $D 2 "--serial"
done
rm -rf $D
