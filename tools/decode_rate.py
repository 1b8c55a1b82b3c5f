#!/usr/bin/env python3
"""The transform kernel on batches of any size (bench.py is fixed to BASELINE's configurations):
    python tools/decode_rate.py [--streams 4096] [--frames 16] [--mix] [--steps 50]
One plan, relaunched; HIP events on the launch stream; two sets of buffers.  Prints one JSON line."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu  # noqa: E402
import aacgpu_workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--mix", action="store_true")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--layout", default="cpe", help="comma-separated elements of a frame, e.g. cpe,cpe,cpe,sce")
    a = ap.parse_args()
    import torch
    layout = tuple(a.layout.split(","))
    wl = aacgpu_workload.make_batch(n_streams=a.streams, n_frames=a.frames, mix=a.mix, layout=layout)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=a.streams, max_channels=max(2, wl["C"]))
    plan = eng.plan(wl["units"])
    dev = torch.device("cuda:0")
    bufs = [(torch.from_numpy(wl["q"]).to(dev), torch.from_numpy(wl["meta"].view(np.int16)).to(dev),
             torch.empty(wl["n_pcm"], dtype=torch.float32, device=dev)) for _ in range(2)]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    go = lambda i: eng.decode_device(plan, bufs[i & 1][0].data_ptr(), bufs[i & 1][1].data_ptr(), bufs[i & 1][2].data_ptr(), side.cuda_stream)
    for i in range(10):
        go(i)
    side.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    for i in range(a.steps):
        go(i)
    e1.record(side)
    side.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    n = a.streams * a.frames
    print(json.dumps({"layout": a.layout, "frames": n, "us_per_launch": ms * 1e3, "frames_per_s": n / (ms * 1e-3), "ns_per_frame": ms * 1e6 / n,
                      "GBps": (wl["q"].nbytes + wl["meta"].nbytes + wl["n_pcm"] * 4) / (ms * 1e-3) / 1e9}))


if __name__ == "__main__":
    main()
