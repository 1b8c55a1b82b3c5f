mkdir -p gpurun_out/r3e
timeout 300 bash tools/ab.sh "" addr.so xpose2.so 2>&1 | tee gpurun_out/r3e/ab_cfg2.log
timeout 300 bash tools/ab.sh "--workload cfg3" addr.so xpose2.so 2>&1 | tee gpurun_out/r3e/ab_cfg3.log
timeout 300 bash tools/ab.sh "--workload cfg5" addr.so xpose2.so 2>&1 | tee gpurun_out/r3e/ab_cfg5.log
timeout 200 bash tools/pmc_insts.sh "" xpose2.so 2>&1 | tee gpurun_out/r3e/pmc.log
