#!/usr/bin/env python3
"""Bytes -> PCM with both stages on the device: aacg_parse_device -> (unit records to the host, plan) -> aacg_decode_device.

    python tools/pipeline_rate.py [--streams 4096] [--frames 16] [--steps 10]

Streams: the 600 stereo frames of tests/js/parse_cases.js (standard codebooks), dealt out as --streams streams of --frames
consecutive frames.  Bit streams and PCM stay in HBM; what crosses PCIe per batch is the 64-byte unit record and the
8-byte result per frame (down) and the planner's run table (up).  Wall-clock per stage, one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--uniform", action="store_true", help="only the frames within 10 %% of the median length (a constant-bit-rate stream)")
    ap.add_argument("--resident", action="store_true", help="one plan for all batches, its unit records refreshed on the device: no host work per batch")
    ap.add_argument("--overlap", action="store_true", help="two batches in flight: the host plans batch N while the GPU parses batch N+1")
    a = ap.parse_args()
    import torch
    d = tempfile.mkdtemp()
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "parse_cases.js"), d, "standard"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    entries = np.fromfile(os.path.join(d, "codebooks.entries"), aacgpu.CODE_ENTRY_DTYPE)
    counts = np.fromfile(os.path.join(d, "codebooks.counts"), np.uint32)
    data = np.fromfile(os.path.join(d, "stereo600.bytes"), np.uint8)
    frames = np.fromfile(os.path.join(d, "stereo600.frames"), aacgpu.PARSE_FRAME_DTYPE)
    if a.uniform:
        med = float(np.median(frames["byte_length"]))
        frames = frames[(frames["byte_length"] > 0.9 * med) & (frames["byte_length"] < 1.1 * med)]
    n = a.streams * a.frames
    reps = (n + len(frames) - 1) // len(frames)
    one = np.concatenate([data, np.zeros((-len(data)) % 16, np.uint8)])
    big = np.concatenate([np.tile(one, reps), np.zeros(32, np.uint8)])
    table = np.tile(frames, reps)
    table["byte_offset"] += np.repeat(np.arange(reps, dtype=np.uint32) * len(one), len(frames))
    table = table[:n]                                   # frame i belongs to stream i // frames, position i % frames
    dev = torch.device("cuda:0")
    t = lambda arr: torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev)
    d_bytes, d_frames = t(big), t(table)
    d_units = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_q = torch.zeros(n * 2 * 1024, dtype=torch.int16, device=dev)
    d_meta = torch.zeros(n * 2 * 120, dtype=torch.int16, device=dev)
    d_res = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
    d_pcm = torch.zeros(n * 2048, dtype=torch.float32, device=dev)
    h_units = torch.zeros(n * 64, dtype=torch.uint8).pin_memory()
    h_res = torch.zeros(n * 8, dtype=torch.uint8).pin_memory()
    p = aacgpu.Parser(entries, counts, sample_index=3)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=a.streams, max_channels=2)
    side = torch.cuda.Stream()
    stream_ids = np.repeat(np.arange(a.streams, dtype=np.uint32), a.frames)
    pcm_off = np.arange(n, dtype=np.uint32) * 2048
    stage = {"parse": 0.0, "records": 0.0, "plan": 0.0, "decode": 0.0}

    def step():
        t0 = time.perf_counter()
        p.parse_device(d_bytes.data_ptr(), d_frames.data_ptr(), n, 1, 2, aacgpu.PARSE_REFERENCE_QUIRKS,
                       d_units.data_ptr(), d_q.data_ptr(), d_meta.data_ptr(), None, d_res.data_ptr(), side.cuda_stream)
        side.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(side):
            h_units.copy_(d_units, non_blocking=True)
            h_res.copy_(d_res, non_blocking=True)
        side.synchronize()
        res = h_res.numpy().view(aacgpu.PARSE_RESULT_DTYPE)
        assert not res["status"].any()
        units = h_units.numpy().view(aacgpu.UNIT_DTYPE)
        units["stream"] = stream_ids
        units["pcm_offset"] = pcm_off
        units["n_out_ch"] = 2
        units["reserved0"] = 0
        t2 = time.perf_counter()
        plan = eng.plan(units)
        t3 = time.perf_counter()
        eng.decode_device(plan, d_q.data_ptr(), d_meta.data_ptr(), d_pcm.data_ptr(), side.cuda_stream)
        side.synchronize()
        t4 = time.perf_counter()
        plan.destroy()
        return t1 - t0, t2 - t1, t3 - t2, t4 - t3

    step()
    for _ in range(a.steps):
        for k, v in zip(stage, step()):
            stage[k] += v / a.steps
    total = sum(stage.values())
    pcm = d_pcm.cpu().numpy()
    assert np.isfinite(pcm).all() and float(np.abs(pcm).max()) > 1e-3
    out = {"frames": n, "streams": a.streams, "ms": {k: round(v * 1e3, 3) for k, v in stage.items()}, "ms_total": round(total * 1e3, 3),
           "frames_per_s": n / total}
    if a.resident:
        # The plan is built once from the streams' structure; every batch is three asynchronous launches on one stream.
        skel = np.zeros(n, aacgpu.UNIT_DTYPE)
        skel["stream"], skel["pcm_offset"], skel["n_out_ch"], skel["n_ch"] = stream_ids, pcm_off, 2, 2
        skel["coef_offset"] = skel["meta_offset"] = np.arange(n, dtype=np.uint32) * 2
        skel["ch"]["group_count"] = 1
        skel["ch"]["group_len"][..., 0] = 1
        eng3 = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=a.streams, max_channels=2)
        plan = eng3.plan(skel)
        d_refused = torch.zeros(1, dtype=torch.int32, device=dev)

        def batch():
            p.parse_device(d_bytes.data_ptr(), d_frames.data_ptr(), n, 1, 2, aacgpu.PARSE_REFERENCE_QUIRKS | aacgpu.PARSE_SKIP_ZERO_FILL,
                           d_units.data_ptr(), d_q.data_ptr(), d_meta.data_ptr(), None, d_res.data_ptr(), side.cuda_stream)
            eng3.plan_refresh_from_parse(plan, d_units.data_ptr(), d_res.data_ptr(), 1, d_refused.data_ptr(), side.cuda_stream)
            eng3.decode_device(plan, d_q.data_ptr(), d_meta.data_ptr(), d_pcm.data_ptr(), side.cuda_stream)

        for _ in range(3):
            batch()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(side)
        for _ in range(a.steps):
            batch()
        e1.record(side)
        side.synchronize()
        wall = (time.perf_counter() - t0) / a.steps
        assert int(d_refused.cpu()[0]) == 0 and np.isfinite(d_pcm.cpu().numpy()).all()
        ms = e0.elapsed_time(e1) / a.steps
        out["resident"] = {"ms_per_batch": round(ms, 3), "wall_ms_per_batch": round(wall * 1e3, 3), "frames_per_s": n / (ms * 1e-3)}
    if a.overlap:
        # Two slots of device / pinned buffers and two streams.  The second slot's streams are numbered after the first's,
        # so the two batches are independent for the engine (consecutive batches of the SAME streams must decode in order).
        eng2 = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=2 * a.streams, max_channels=2)
        slots = []
        for k in range(2):
            slots.append({"units": torch.zeros(n * 64, dtype=torch.uint8, device=dev), "q": torch.zeros(n * 2 * 1024, dtype=torch.int16, device=dev),
                          "meta": torch.zeros(n * 2 * 120, dtype=torch.int16, device=dev), "res": torch.zeros(n * 8, dtype=torch.uint8, device=dev),
                          "pcm": torch.zeros(n * 2048, dtype=torch.float32, device=dev), "h_units": torch.zeros(n * 64, dtype=torch.uint8).pin_memory(),
                          "h_res": torch.zeros(n * 8, dtype=torch.uint8).pin_memory(), "stream": torch.cuda.Stream(), "ids": stream_ids + k * a.streams,
                          "plan": None})

        def launch_parse(sl):
            p.parse_device(d_bytes.data_ptr(), d_frames.data_ptr(), n, 1, 2, aacgpu.PARSE_REFERENCE_QUIRKS, sl["units"].data_ptr(), sl["q"].data_ptr(),
                           sl["meta"].data_ptr(), None, sl["res"].data_ptr(), sl["stream"].cuda_stream)
            with torch.cuda.stream(sl["stream"]):
                sl["h_units"].copy_(sl["units"], non_blocking=True)
                sl["h_res"].copy_(sl["res"], non_blocking=True)

        def finish(sl):
            sl["stream"].synchronize()                      # parse + records of this slot
            assert not sl["h_res"].numpy().view(aacgpu.PARSE_RESULT_DTYPE)["status"].any()
            units = sl["h_units"].numpy().view(aacgpu.UNIT_DTYPE)
            units["stream"] = sl["ids"]
            units["pcm_offset"] = pcm_off
            units["n_out_ch"] = 2
            units["reserved0"] = 0
            if sl["plan"] is not None:
                sl["plan"].destroy()
            sl["plan"] = eng2.plan(units)
            eng2.decode_device(sl["plan"], sl["q"].data_ptr(), sl["meta"].data_ptr(), sl["pcm"].data_ptr(), sl["stream"].cuda_stream)

        launch_parse(slots[0])
        for i in range(4):                                  # warm-up
            launch_parse(slots[(i + 1) & 1])
            finish(slots[i & 1])
        torch.cuda.synchronize()
        launch_parse(slots[0])
        t0 = time.perf_counter()
        for i in range(2 * a.steps):
            launch_parse(slots[(i + 1) & 1])
            finish(slots[i & 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (2 * a.steps)
        assert np.isfinite(slots[0]["pcm"].cpu().numpy()).all()
        out["overlapped"] = {"ms_per_batch": round(dt * 1e3, 3), "frames_per_s": n / dt}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
