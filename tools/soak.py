#!/usr/bin/env python3
"""Soak run on a GPU box (not a pytest): random batches of random shape against the oracle for a few minutes —
layouts, sequences, shapes, groupings, band types, chain lengths 1..70 (all run kinds, incl. double duty),
both seams, engines reused across batches so that overlap state and plan parity are exercised.
Usage: python tools/soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, aacgpu, aacgpu_workload as W, orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
o = orc.load()
rng = np.random.default_rng(20261001)
t0 = time.time(); n_batches = n_frames = n_narrow = n_piped = 0; worst = 0.0
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
    # a third of the batches exercise the optional modes: TNS SPEC (encoder-like filters), PNS SPEC, or both
    mode = int(rng.integers(0, 6))
    use_tns, use_pns = mode in (1, 3), mode in (2, 3)
    units, meta, tns = wl["units"], wl["meta"], None
    if use_pns:
        units, meta = W.add_pns(dict(units=units, meta=meta), seed=seed)
    if use_tns:
        units, tns = W.add_tns(dict(units=units), seed=seed)
    wl = dict(wl, units=units, meta=meta)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C, tns_mode=int(use_tns), pns_mode=int(use_pns))
    engf = aacgpu.Engine(aacgpu.INPUT_SPEC_F32, S, C, tns_mode=int(use_tns))
    # half of the engines take the old route for chains longer than a run (a recomputed frame per later run instead of the rendezvous)
    narrow = bool(rng.integers(0, 2))
    if narrow:
        eng.debug_set_route(aacgpu.DEBUG_ROUTE_RECOMPUTE); engf.debug_set_route(aacgpu.DEBUG_ROUTE_RECOMPUTE)
        n_narrow += 1
    for rep in range(int(rng.integers(1, 4))):                 # consecutive batches of the same streams
        ov_in = ov.copy()
        ref, spec = o.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, want_spec=True, tns=tns, pns=use_pns)
        if use_tns:                                            # spec_out is post-TNS: the f32 seam needs the pre-TNS spectrum
            _, spec = o.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov_in.copy(), want_spec=True, pns=use_pns)
        sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-12
        for e, x, m in ((eng, wl["q"], meta), (engf, spec, None)):
            pcm = e.decode_batch(units, x, m, wl["n_pcm"], tns=tns)
            err = float(np.sqrt(np.mean((pcm.astype(np.float64) - ref) ** 2))) / sig
            worst = max(worst, err)
            # TNS SPEC: block scan vs serial evaluation differ by rounding x filter gain; a single frame (and with it the
            # overlap state) can be 1e-5 off where the batch average is 1e-6 (tests/test_tns_spec.py)
            assert np.array_equal(np.isnan(pcm), np.isnan(ref)) and err < (2e-5 if use_tns else 5e-6), (seed, rep, mode, err)
        got = np.stack([[eng.get_overlap(s, c) for c in range(C)] for s in range(S)])
        ov_err = float(np.sqrt(np.mean((got.astype(np.float64) - ov) ** 2))) / (float(np.sqrt(np.mean(ov.astype(np.float64) ** 2))) + 1e-12)
        # a single tail (the last frame only) is noisier than the batch average: 5e-6 is seen once in ~50 k batches
        assert ov_err < (1e-4 if use_tns else 2e-5), (seed, rep, mode, "overlap", ov_err)
        n_batches += 1; n_frames += len(wl["units"])
    # a third of the time: the same batch as a PLAN, launched 3..14 times in a row through aacg_decode_pipelined (launches
    # overlap, their chains meet in rendezvous cells; TNS / PNS batches take the in-run stage kernels' rendezvous builds),
    # every launch against the oracle continued from launch to launch
    if rng.random() < 0.34:
        engp = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C, tns_mode=int(use_tns), pns_mode=int(use_pns))
        plan = engp.plan(units, tns=tns)
        n = int(rng.integers(3, 15))
        dq = torch.from_numpy(np.ascontiguousarray(wl["q"])).cuda(); dm = torch.from_numpy(np.ascontiguousarray(meta).view(np.int16)).cuda()
        outs = [torch.full((wl["n_pcm"],), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        for j in range(n):
            engp.decode_pipelined(plan, dq.data_ptr(), dm.data_ptr(), outs[j].data_ptr())
        engp.synchronize(); torch.cuda.synchronize()
        ovp = np.zeros((S, C, 1024), np.float32)
        for j in range(n):
            ref = o.decode_batch(units, wl["q"], meta, wl["n_pcm"], ovp, tns=tns, pns=use_pns)
            pcm = outs[j].cpu().numpy()
            sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-12
            err = float(np.sqrt(np.mean((pcm.astype(np.float64) - ref) ** 2))) / sig
            worst = max(worst, err)
            assert np.array_equal(np.isnan(pcm), np.isnan(ref)) and err < (2e-5 if use_tns else 5e-6), (seed, "pipelined", j, mode, err)
        plan.destroy(); engp.close()
        n_piped += n
    eng.close(); engf.close()
print("soak ok: %d batches, %d units, worst relative rms error %.2e, %.0f s; %d engine pairs on the recompute route for long chains; %d launches through the pipeline" % (n_batches, n_frames, worst, time.time() - t0, n_narrow, n_piped))
