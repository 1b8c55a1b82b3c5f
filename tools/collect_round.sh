#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench lines and rocprofv3 summaries into gpurun_out/<tag>/.
# usage: tools/collect_round.sh r05
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 200 python3 -c "import torch; torch.zeros(1).cuda()" >/dev/null 2>&1 || { echo "gpu init failed or slow"; exit 1; }
b() { name=$1; shift; python3 bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "bench $name failed" >> $OUT/failures.txt; }
# the headline and its variants: launches through aacg_decode_pipelined (the default), and --serial = rounds 1-4's method
b quant
b quant_20steps --steps 20 --warmup 5
b quant_serial --serial --no-cpu-baseline
b quant_serial_20steps --serial --steps 20 --warmup 5 --no-cpu-baseline
b spec --input spec --no-cpu-baseline
b spec_serial --input spec --serial --no-cpu-baseline
b cfg3 --workload cfg3 --no-cpu-baseline
b cfg3_serial --workload cfg3 --serial --no-cpu-baseline
b cfg4 --workload cfg4 --no-cpu-baseline
b cfg4_serial --workload cfg4 --serial --no-cpu-baseline
b cfg5 --workload cfg5 --steps 1000 --warmup 200
b cfg5_serial --workload cfg5 --serial --steps 1000 --warmup 200 --no-cpu-baseline
b cfg5_spec --workload cfg5 --input spec --steps 1000 --warmup 200 --no-cpu-baseline
b cfg3_tns_spec_quant --workload cfg3 --tns spec --steps 1000 --warmup 200 --no-cpu-baseline
b cfg3_tns_spec_f32 --workload cfg3 --tns spec --input spec --steps 1000 --warmup 200 --no-cpu-baseline
b cfg3_tns_spec_quant_serial --workload cfg3 --tns spec --serial --steps 1000 --warmup 200 --no-cpu-baseline
b cfg5_cce_spec --workload cfg5 --cce spec --steps 500 --warmup 100 --no-cpu-baseline
b quant_i16out --output i16 --no-cpu-baseline
b quant_i16out_serial --output i16 --serial --no-cpu-baseline
b quant_pipelines2 --pipelines 2 --no-cpu-baseline
b quant_2ranks_shared_gpu --gpus 2 --dist-backend gloo --share-gpu --steps 1000 --warmup 200
# the driver's launch line with one rank: RCCL carries the barrier and the 8-byte reductions
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline \
  > $OUT/bench_quant_torchrun_rccl_1rank.json 2> $OUT/bench_quant_torchrun_rccl_1rank.err || echo "bench torchrun failed" >> $OUT/failures.txt
# the plugin surface: 256 streams through readChunk() on one JavaScript thread (independent decoders, SharedEngine, resident SharedEngine);
# and the same with 5.1 streams (SCE + CPE + CPE + LFE: config 5's shape behind the plugin surface)
node tools/readchunk_rate.js --streams 256 100 > $OUT/readchunk_256streams.json 2> $OUT/readchunk_256streams.err || echo "readchunk failed" >> $OUT/failures.txt
node tools/readchunk_rate.js --file surround48 --streams 256 100 > $OUT/readchunk_256streams_surround.json 2> $OUT/readchunk_256streams_surround.err || echo "readchunk surround failed" >> $OUT/failures.txt
# bytes -> PCM through the C ABI (aacg_pipeline_*): one batch at a time, and with batches in flight
( A=tests/golden/streams/stereo48.aac
  tools/micro/resident_drive $A --sync; for l in 2 3 4 5 6 8; do tools/micro/resident_drive $A --lanes $l; done
  tools/micro/resident_drive $A --sync --i16; for l in 4 5 6 8; do tools/micro/resident_drive $A --lanes $l --i16; done; tools/micro/resident_drive $A --lanes 5 --pageable --batches 50
  tools/micro/resident_drive tests/golden/streams/surround48.aac --lanes 5 --streams 256; tools/micro/resident_drive tests/golden/streams/mono22.aac --lanes 5 ) > $OUT/resident.jsonl 2> $OUT/resident.err
# where a resident batch's time goes: kernel + copy trace of the same driver (f32 and int16 PCM), tools/resident_budget.py
( for v in "" "--i16"; do
    rm -rf $OUT/prof_resident; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/prof_resident -o resident -- tools/micro/resident_drive tests/golden/streams/stereo48.aac --lanes 5 --batches 50 $v > $OUT/prof_resident.json 2> $OUT/prof_resident.err
    python3 tools/resident_budget.py $(find $OUT/prof_resident -name "resident_kernel_trace.csv") $(find $OUT/prof_resident -name "resident_memory_copy_trace.csv") "5 lanes, ${v:-f32} PCM, traced: $(python3 -c "import json,sys; print('%.3f ms per batch by the driver' % json.loads(open('$OUT/prof_resident.json').read().strip().splitlines()[-1])['ms_per_batch_median'])")"
  done; rm -rf $OUT/prof_resident ) > $OUT/resident_budget.txt 2>&1
# what the headline's launch would cost if the dequantisation (or the PCM stores) were free: the profile build's skipping switches
bash tools/floor.sh 2 > $OUT/dequant_floor.txt 2>&1
# the headline route from a tight C loop, how the host waits (aacg_wait.h), and an event on every launch
( for m in 0 1 2 3; do echo "wait mode $m: $(tools/micro/pipe_drive --wait-mode $m --repeats 3 2>/dev/null)"; done
  echo "default policy: $(tools/micro/pipe_drive --repeats 3 2>/dev/null)"
  echo "an event bound to every launch: $(tools/micro/pipe_drive --mark-all --repeats 3 2>/dev/null)"
  echo "launch behind launch: $(tools/micro/pipe_drive --serial --repeats 3 2>/dev/null)" ) > $OUT/wait_modes.txt 2>&1
# microbenchmarks behind the pipeline's design choices
( for m in launch_cost queue_map stop_event; do echo "== tools/micro/$m"; timeout 120 tools/micro/$m 2>&1; done ) > $OUT/micro.txt
# rocprofv3: kernel trace + PMC passes (tools/prof.sh); pipelined launches overlap, so the trace's row spacing is printed beside the stats
bash tools/prof.sh $TAG/prof_quant > $OUT/prof_quant.log 2>&1
# the headline's kernel trace from the C driver (tools/micro/pipe_drive: no interpreter between two launches), beside bench.py's own
bash tools/prof_pipe.sh $TAG/prof_pipe > $OUT/prof_pipe.log 2>&1
cat $OUT/prof_pipe/pipe_drive.json $OUT/prof_pipe/pipe_drive_traced.json > $OUT/pipe_drive.jsonl 2>/dev/null
bash tools/prof.sh $TAG/prof_quant_serial --serial > $OUT/prof_quant_serial.log 2>&1
bash tools/prof.sh $TAG/prof_spec --input spec > $OUT/prof_spec.log 2>&1
bash tools/prof.sh $TAG/prof_cfg5 --workload cfg5 > $OUT/prof_cfg5.log 2>&1
bash tools/prof.sh $TAG/prof_cfg3_tns --workload cfg3 --tns spec > $OUT/prof_cfg3_tns.log 2>&1
# keep what is judged: summaries, kernel stats, dispatch spacing (the raw rocprofv3 trees stay behind)
for p in prof_quant prof_quant_serial prof_spec prof_cfg5 prof_cfg3_tns; do
  cp $OUT/$p/summary.txt $OUT/${p}_summary.txt 2>/dev/null
  cp $OUT/$p/intervals.txt $OUT/${p}_intervals.txt 2>/dev/null
  find $OUT/$p/trace -name "*kernel_stats.csv" -exec cp {} $OUT/${p}_kernel_stats.csv \; 2>/dev/null
  rm -rf $OUT/$p
done
# the headline's intervals and kernel stats: the C driver's trace (bench.py's stay beside them)
if [ -s $OUT/prof_pipe/intervals.txt ]; then
  mv $OUT/prof_quant_intervals.txt $OUT/prof_quant_bench_intervals.txt; mv $OUT/prof_quant_kernel_stats.csv $OUT/prof_quant_bench_kernel_stats.csv
  ( cat $OUT/prof_pipe/intervals.txt
    python3 - <<PY
import json
u = json.loads(open("$OUT/prof_pipe/pipe_drive.json").read().strip().splitlines()[-1]); t = json.loads(open("$OUT/prof_pipe/pipe_drive_traced.json").read().strip().splitlines()[-1])
print("driver: tools/micro/pipe_drive (C, aacg_decode_pipelined in a tight loop); the same binary untraced: %.2f us per launch by its dispatches' events (%.2f by the host's clock); under the tracer it measures %.2f itself" % (u["us_per_launch_events"], u["us_per_launch_host_clock"], t["us_per_launch_events"]))
for n in ("default", "20steps"):
    try:
        d = json.loads(open("$OUT/prof_pipe/bench_%s.json" % n).read().strip().splitlines()[-1])
        print("driver: bench.py (%s) on the same box, untraced: %.2f us per launch" % (n, d["ms_per_step"] * 1e3))
    except Exception as e:
        pass
PY
  ) > $OUT/prof_quant_intervals.txt
  cp $OUT/prof_pipe/kernel_stats.csv $OUT/prof_quant_kernel_stats.csv
fi
rm -rf $OUT/prof_pipe/trace
# per-wave phase timelines (profile build): launch after launch, and pipelined steady state
if [ -f aac.js_amd/csrc/variants/profile.so ]; then
  ( echo "== serial launches (aacg_decode_device): one launch, every CU starts together"; AACGPU_LIB=aac.js_amd/csrc/variants/profile.so AACG_ABLATE=16 timeout 120 python3 tools/timeline.py quant 2>/dev/null
    echo; echo "== pipelined launches (aacg_decode_pipelined): the fourth launch from the end of 3001, its neighbours running beside it"; AACGPU_LIB=aac.js_amd/csrc/variants/profile.so AACG_ABLATE=16 TL_PIPE=1 timeout 120 python3 tools/timeline.py quant 2>/dev/null ) > $OUT/timeline.txt
fi
ls -la $OUT
