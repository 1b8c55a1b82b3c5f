#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path (never the headline metric: DESIGN.md §5).
Pinned buffers, two batches in flight via aacg_submit / aacg_wait, BASELINE config 2 batches.
    python tools/pcie_rate.py [i16]      i16: an AACG_OUTPUT_I16 engine (half the bytes back over PCIe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import numpy as np, aacgpu, aacgpu_workload
S, T, NB, ROUNDS = 256, 16, 4, 40
I16 = len(sys.argv) > 1 and sys.argv[1] == "i16"
eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, 2, output_kind=aacgpu.OUTPUT_I16 if I16 else aacgpu.OUTPUT_F32)
wl = aacgpu_workload.make_batch(S, T)
sets = []
for b in range(NB):
    q = eng.pinned(wl["q"].shape, np.int16); q[:] = wl["q"]
    m = eng.pinned(wl["meta"].shape, np.uint16); m[:] = wl["meta"]
    pcm = eng.pinned((wl["n_pcm"],), np.int16 if I16 else np.float32)
    sets.append((q, m, pcm))
units = wl["units"]
def run(n):
    tickets = []
    for i in range(n):
        q, m, pcm = sets[i % NB]
        tickets.append(eng.submit(units, q, m, pcm))
        if i >= 1: eng.wait(tickets[i - 1])
    eng.wait(tickets[-1])
run(6)
t0 = time.perf_counter(); run(ROUNDS); dt = time.perf_counter() - t0
fps = S * T * ROUNDS / dt
print("host-buffer path (%s PCM), pinned, 2 in flight: %.3f ms per 4096-frame batch, %.2f M stereo frames/s (%.0fx real time), "
      "%.1f GB/s H2D + %.1f GB/s D2H" % ("int16" if I16 else "f32", dt / ROUNDS * 1e3, fps / 1e6, fps / 46.875,
      (wl["q"].nbytes + wl["meta"].nbytes) * ROUNDS / dt / 1e9, wl["n_pcm"] * (2 if I16 else 4) * ROUNDS / dt / 1e9))
# synchronous, pageable: a caller that reuses its arrays (staged through the engine's pinned buffers), and one that takes a
# fresh output array per call (the page faults of 33 MB of new memory are then part of the call)
qn, mn = wl["q"].copy(), wl["meta"].copy()
out = np.empty(wl["n_pcm"], np.int16 if I16 else np.float32)
for i in range(3): eng.decode_batch(units, qn, mn, wl["n_pcm"], out=out)
t0 = time.perf_counter()
for i in range(10): eng.decode_batch(units, qn, mn, wl["n_pcm"], out=out)
dt = time.perf_counter() - t0
print("synchronous aacg_decode_batch, pageable memory, arrays reused: %.3f ms per batch, %.2f M stereo frames/s" % (dt / 10 * 1e3, S * T * 10 / dt / 1e6))
t0 = time.perf_counter()
for i in range(10): eng.decode_batch(units, qn, mn, wl["n_pcm"])
dt = time.perf_counter() - t0
print("synchronous aacg_decode_batch, pageable memory, fresh output array per call: %.3f ms per batch, %.2f M stereo frames/s" % (dt / 10 * 1e3, S * T * 10 / dt / 1e6))
# the host's share of a device-resident batch: a plan per batch against one plan refreshed from the host's unit records
t0 = time.perf_counter()
for i in range(20):
    p = eng.plan(units); p.destroy()
t_plan = (time.perf_counter() - t0) / 20
p = eng.plan(units)
t0 = time.perf_counter()
for i in range(20): eng.plan_refresh_units(p, units)
eng.synchronize()
t_refresh = (time.perf_counter() - t0) / 20
p.destroy()
print("host work per device-resident batch of %d units: aacg_plan_create + destroy %.0f us, aacg_plan_refresh_units %.0f us" % (len(units), t_plan * 1e6, t_refresh * 1e6))
