mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or tables or spectral_stage or scenarios or fuzz" 2>&1 | tail -5 | tee gpurun_out/r3d/kat.log
bash tools/ab.sh "" mirror.so addr.so 2>&1 | tee gpurun_out/r3d/ab_cfg2.log
bash tools/ab.sh "--input spec" r2.so addr.so 2>&1 | tee gpurun_out/r3d/ab_spec.log
bash tools/pmc_insts.sh "" addr.so 2>&1 | tee gpurun_out/r3d/pmc.log
