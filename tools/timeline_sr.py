#!/usr/bin/env python3
"""Per-wave, per-frame phase timeline of the stream-resident kernel (profiling aid).
Needs the profile build and AACG_ABLATE=16:
    make -C aac.js_amd/csrc profile
    AACGPU_LIB=aac.js_amd/csrc/variants/profile.so AACG_ABLATE=16 python tools/timeline_sr.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import numpy as np, torch, aacgpu, aacgpu_workload

S, T = int(os.environ.get("TL_STREAMS", "256")), int(os.environ.get("TL_FRAMES", "16"))
layout = ("cpe", "cpe", "cpe", "sce")
eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, 7)
wl = aacgpu_workload.make_batch(S, T, layout=layout, mix=True)
plan = eng.plan(wl["units"])
d_in = torch.from_numpy(wl["q"]).cuda(); d_meta = torch.from_numpy(wl["meta"].view(np.int16)).cuda()
d_out = torch.empty(wl["n_pcm"], dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(5):
    eng.decode_device(plan, d_in.data_ptr(), d_meta.data_ptr(), d_out.data_ptr(), 0)
eng.synchronize()
raw = np.zeros(1 << 20, np.float32)
eng._check(eng.lib.aacg_get_table(eng.handle, 100, raw.ctypes.data, raw.size))
K = (T + 3) // 4
t = raw.view(np.uint64)[: S * 16 * 8 * 8].reshape(S, 16, 8, 8)[:, :, :K].astype(np.float64) * 0.01     # 100 MHz ticks -> us
t0 = t[:, :, 0, 0].min()
names = ["start", "slot free", "imdct done", "prev tails", "stage turn", "all staged", "stored"]
print("median over workgroups, us since the first wave's start; wave = ring * 4 + element; frame = ring + 4 k")
for w in range(16):
    for k in range(K):
        row = [np.median(t[:, w, k, i] - t0) for i in range(7)]
        print("wave %2d frame %2d: " % (w, w // 4 + 4 * k) + "  ".join("%s %7.2f" % (names[i], row[i]) for i in range(7)))
d = lambda a, b: np.median(t[:, :, :, b] - t[:, :, :, a])
print("phase medians: start->slot free %.2f, ->imdct done %.2f, ->prev tails %.2f, ->stage turn %.2f, ->all staged %.2f, ->stored %.2f" %
      (d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6)))
print("kernel span %.2f us" % (t[:, :, :, 6].max() - t0))
