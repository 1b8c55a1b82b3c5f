#!/usr/bin/env python3
"""Rate of the device front end (aacg_parse_frames) and of bytes -> PCM with both stages on the device.

    python tools/parse_rate.py [--frames 65536] [--steps 50] [--standin]

Streams: the 600 stereo 48 kHz frames of tests/js/parse_cases.js (synthetic writer; the standard codebooks unless --standin),
repeated to --frames.  Timed with HIP events on the launch stream, inputs
resident in HBM.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--standin", action="store_true")
    ap.add_argument("--uniform", action="store_true", help="only the frames within 10 %% of the median length (a constant-bit-rate stream)")
    ap.add_argument("--decode", action="store_true", help="also time parse + plan-reuse decode of the same batch")
    a = ap.parse_args()
    import torch
    d = tempfile.mkdtemp()
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "parse_cases.js"), d, "synthetic" if a.standin else "standard"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "SKIP" not in r.stdout, r.stdout + r.stderr
    entries = np.fromfile(os.path.join(d, "codebooks.entries"), aacgpu.CODE_ENTRY_DTYPE)
    counts = np.fromfile(os.path.join(d, "codebooks.counts"), np.uint32)
    data = np.fromfile(os.path.join(d, "stereo600.bytes"), np.uint8)
    frames = np.fromfile(os.path.join(d, "stereo600.frames"), aacgpu.PARSE_FRAME_DTYPE)
    if a.uniform:
        med = float(np.median(frames["byte_length"]))
        frames = frames[(frames["byte_length"] > 0.9 * med) & (frames["byte_length"] < 1.1 * med)]
    reps = (a.frames + len(frames) - 1) // len(frames)
    pad = (-len(data)) % 16
    one = np.concatenate([data, np.zeros(pad, np.uint8)])
    big = np.concatenate([np.tile(one, reps), np.zeros(32, np.uint8)])
    table = np.tile(frames, reps)
    table["byte_offset"] += np.repeat(np.arange(reps, dtype=np.uint32) * len(one), len(frames))
    table = table[:a.frames]
    n = len(table)
    dev = torch.device("cuda:0")
    t = lambda arr: torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev)
    d_bytes, d_frames = t(big), t(table)
    d_units = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_q = torch.zeros(n * 2 * 1024, dtype=torch.int16, device=dev)
    d_meta = torch.zeros(n * 2 * 120, dtype=torch.int16, device=dev)
    d_res = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
    p = aacgpu.Parser(entries, counts, sample_index=3)
    side = torch.cuda.Stream()                          # a real stream handle: 0 would select the parser's own stream
    stream = side.cuda_stream
    torch.cuda.synchronize()
    go = lambda: p.parse_device(d_bytes.data_ptr(), d_frames.data_ptr(), n, 1, 2, aacgpu.PARSE_REFERENCE_QUIRKS,
                                d_units.data_ptr(), d_q.data_ptr(), d_meta.data_ptr(), None, d_res.data_ptr(), stream)
    for _ in range(a.warmup):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    for _ in range(a.steps):
        go()
    e1.record(side)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    res = d_res.cpu().numpy().view(aacgpu.PARSE_RESULT_DTYPE)
    assert not res["status"].any()
    in_bytes = int(table["byte_length"].sum())
    out = {"kernel": "aacg_parse_frames", "frames": n, "us_per_batch": ms * 1e3, "frames_per_s": n / (ms * 1e-3),
           "bytes_per_frame": in_bytes / n, "stream_GBps": in_bytes / (ms * 1e-3) / 1e9,
           "written_GBps": n * (2 * 2048 + 2 * 240 + 64 + 8) / (ms * 1e-3) / 1e9, "codebooks": "stand-in" if a.standin else "standard"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
