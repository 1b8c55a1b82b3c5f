#!/bin/bash
# resource report of the one-channel-per-wave kernels (from anywhere)
cd /root/repo/aac.js_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} "$@" -Rpass-analysis=kernel-resource-usage -c aacg_engine8.hip -o /dev/null 2>&1 | grep -E "error|Function Name|VGPRs:|Scratch|VGPRs Spill|SGPRs Spill|Occupancy"
