#!/bin/bash
# Builds tests/emu's kernel source with -fsanitize=address,undefined and runs tools/asan_emu.py against it (CPU only).
set -e
R=$(cd $(dirname $0)/.. && pwd)
cd $R/tests/emu
g++ -O1 -g -std=c++17 -fPIC -fno-strict-aliasing -DAACG_EMU_BUILD -I. -pthread -Wno-unused-function -Wno-unknown-pragmas \
    -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize=vptr -shared -o /tmp/libaacg_emu_asan.so \
    emu_lib.cpp ../../aac.js_amd/csrc/aacg_tables.cpp ../../aac.js_amd/csrc/aacg_plan.cpp ../../aac.js_amd/csrc/aacg_routes.cpp ../../aac.js_amd/csrc/aacg_parse_host.cpp
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 python3 tools/asan_emu.py
