#!/usr/bin/env python3
"""One rank of a sharded decode (started by aacgpu_shard.self_launch / torch.distributed.run): TEST KIT.

Every rank builds the same synthetic batch of --streams x --frames, keeps the units of its own stream shard
(aacgpu_shard.stream_shard), decodes them with --decoder and writes its slice of the PCM buffer to
<out>/pcm_rank<r>.f32.  No data-path collective: the harness carries a barrier, a MAX and a SUM of 8 bytes.

  --decoder engine   the HIP engine on cuda:(LOCAL_RANK mod device_count) — the product path (-m gpu tests)
  --decoder oracle   the CPU checker standing in for a device (CPU tests of the harness itself; no GPU here)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--streams", type=int, default=6)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--decoder", choices=["engine", "oracle"], required=True)
    ap.add_argument("--backend", choices=["gloo", "nccl"], default="gloo")
    ap.add_argument("--shard", choices=["stream", "time"], default="stream",
                    help="time: every rank takes a span of frames of EVERY stream (aacgpu_shard.time_shard) instead of whole streams")
    args = ap.parse_args()
    import aacgpu_shard
    import aacgpu_workload
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dist = aacgpu_shard.init_process_group(args.backend)
    wl = aacgpu_workload.make_batch(n_streams=args.streams, n_frames=args.frames, mix=True, intensity=True, seed=0xAAC00004)
    per_frame = 1024 * wl["C"]
    per_stream = args.frames * per_frame
    t_of = (wl["units"]["pcm_offset"] // per_frame) % args.frames          # frame number inside its stream
    if args.shard == "time":
        lo, hi, warm = aacgpu_shard.time_shard(args.frames, rank, world)
        mine = wl["units"][(t_of >= lo - warm) & (t_of < hi)]
    else:
        lo, hi = aacgpu_shard.stream_shard(args.streams, rank, world)
        mine = wl["units"][(wl["units"]["stream"] >= lo) & (wl["units"]["stream"] < hi)]
    if args.decoder == "engine":
        import torch
        import aacgpu
        device = local % torch.cuda.device_count()
        torch.cuda.set_device(device)
        sync = torch.cuda.synchronize
        eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=args.streams, max_channels=wl["C"], device=device)
        box = {}

        def body():
            box["pcm"] = eng.decode_batch(mine, wl["q"], wl["meta"], wl["n_pcm"])
    else:
        import orc
        sync = lambda: None
        ov = np.zeros((args.streams, wl["C"], 1024), np.float32)
        box = {}

        def body():
            box["pcm"] = orc.load().decode_batch(mine, wl["q"], wl["meta"], wl["n_pcm"], ov)
    t_mine, t_max = aacgpu_shard.timed(dist, sync, body)
    frames = aacgpu_shard.reduce_sum(dist, len(mine))              # decoded, warm-up frames included
    if args.shard == "time":                           # frames [lo, hi) of every stream; the warm-up frame's PCM is dropped
        box["pcm"].reshape(args.streams, args.frames, per_frame)[:, lo:hi].tofile(os.path.join(args.out, "pcm_rank%d.f32" % rank))
    else:
        box["pcm"][lo * per_stream:hi * per_stream].tofile(os.path.join(args.out, "pcm_rank%d.f32" % rank))
    if rank == 0:
        with open(os.path.join(args.out, "summary.json"), "w") as f:
            json.dump({"world": world, "frames": frames, "t_max": t_max, "t_rank0": t_mine}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
