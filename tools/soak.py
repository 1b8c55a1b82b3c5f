#!/usr/bin/env python3
"""Soak run on a GPU box (not a pytest): random batches of random shape against the oracle for a few minutes —
layouts, sequences, shapes, groupings, band types, chain lengths 1..70 (all run kinds, incl. double duty),
both seams, engines reused across batches so that overlap state and plan parity are exercised.
Usage: python tools/soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, aacgpu, aacgpu_workload as W, orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
o = orc.load()
rng = np.random.default_rng(20261001)
t0 = time.time(); n_batches = n_frames = 0; worst = 0.0
while time.time() - t0 < budget:
    seed = int(rng.integers(1, 1 << 30))
    if rng.random() < 0.5:
        wl = W.random_batch(seed, n_streams=int(rng.integers(1, 6)), max_frames=int(rng.integers(1, 71)))
        S, C = wl["n_streams"], wl["max_channels"]
    else:
        layout = [("cpe",), ("sce",), ("cpe", "sce"), ("cpe", "cpe", "cpe", "sce")][int(rng.integers(0, 4))]
        S = int(rng.integers(1, 9))
        wl = W.make_batch(n_streams=S, n_frames=int(rng.integers(1, 71)), layout=layout, mix=True, intensity=bool(rng.integers(0, 2)), seed=seed)
        C = wl["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C)
    engf = aacgpu.Engine(aacgpu.INPUT_SPEC_F32, S, C)
    for rep in range(int(rng.integers(1, 4))):                 # consecutive batches of the same streams
        ref, spec = o.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
        sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-12
        for e, x, m in ((eng, wl["q"], wl["meta"]), (engf, spec, None)):
            pcm = e.decode_batch(wl["units"], x, m, wl["n_pcm"])
            err = float(np.sqrt(np.mean((pcm.astype(np.float64) - ref) ** 2))) / sig
            worst = max(worst, err)
            assert np.array_equal(np.isnan(pcm), np.isnan(ref)) and err < 5e-6, (seed, rep, err)
        got = np.stack([[eng.get_overlap(s, c) for c in range(C)] for s in range(S)])
        assert np.abs(got - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max())), (seed, rep)
        n_batches += 1; n_frames += len(wl["units"])
    eng.close(); engf.close()
print("soak ok: %d batches, %d units, worst relative rms error %.2e, %.0f s" % (n_batches, n_frames, worst, time.time() - t0))
