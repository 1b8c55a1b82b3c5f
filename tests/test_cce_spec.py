"""AACG_CCE_SPEC: coupling channel elements as cce.js:121-158 + decoder.js:406-433 were meant to work.

The reference never couples (`1 === true`, decoder.js:418; coupling point 3, cce.js:69-70; undefined `swb`, cce.js:149):
there is no reference behaviour to pin to, so this mode is PARITY-UNPINNED BY THE REFERENCE.  What is checked instead:

  not gpu : the oracle's orchestration (orc_decode_batch_cce: which element adds to which channel, at which point of
            decoder.js:258-272 / 304-322, with which gain) against an independent numpy form built from the oracle's own
            pinned pieces (dequantisation, filterbank); the kernels' source in the lane emulator against the oracle
  gpu     : the engine against the oracle — all three coupling points, several elements per frame, chains longer than a
            run, consecutive batches (an independently switched element owns overlap state), both seams, together with
            AACG_TNS_SPEC (coupling before / after the filter); error behaviour
"""
import numpy as np
import pytest

import aacgpu
import aacgpu_workload
import emu_lib
import orc

RMS_REL = 5e-6


def workload(layout, points, S=2, T=5, seed=3):
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=seed)
    units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=seed + 100)
    return wl, units, q, meta, cce


def rel(got, ref):
    d = got.astype(np.float64) - ref
    return float(np.sqrt(np.mean(d * d)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2)))


def numpy_coupling(o, wl, units, q, meta, cce, S, H):
    """Independent form: spectra of every unit from the oracle WITHOUT coupling (want_spec on a batch in which the CCEs are
    ordinary single channels), the coupling in numpy float64 -> float32, then the oracle's filterbank channel by channel."""
    C = wl["C"]
    plain = units.copy()
    is_cce = (plain["flags"] & aacgpu.UNIT_CCE) != 0
    # spectra: decode the non-CCE units and, separately, the CCE units as mono elements of a one-channel stream
    ov = np.zeros((S, H, 1024), np.float32)
    _, spec = o.decode_batch(plain[~is_cce], q, meta, wl["n_pcm"], ov, want_spec=True)
    mono = plain[is_cce].copy()
    mono["flags"] = 0
    mono["n_out_ch"] = 1
    mono["channel"] = 0
    mono["pcm_offset"] = np.arange(len(mono)) * 1024
    ov1 = np.zeros((S, 1, 1024), np.float32)
    _, spec_c = o.decode_batch(mono, q, meta, len(mono) * 1024, ov1, want_spec=True)
    spec = spec + spec_c                                         # disjoint blocks
    swb_long, swb_short = o.swb_offsets(3, True), o.swb_offsets(3, False)
    pcm = np.zeros(wl["n_pcm"], np.float32)
    ovl = np.zeros((S, H, 1024), np.float32)
    keys = list(zip(units["stream"].tolist(), units["pcm_offset"].tolist()))
    frames = {}
    for i, k in enumerate(keys):
        frames.setdefault(k, []).append(i)
    for k in sorted(frames, key=lambda k: min(frames[k])):
        idx = frames[k]
        cces = [i for i in idx if units[i]["flags"] & aacgpu.UNIT_CCE]
        time_c = {}
        for i in cces:
            u = units[i]
            if cce[u["reserved1"]]["coupling_point"] == 2:
                info = u["ch"][0]
                time_c[i] = o.filterbank(int(info["window_sequence"]), int(info["window_shape"]), int(info["window_shape_prev"]),
                                         spec[u["coef_offset"]], ovl[u["stream"], u["channel"]])
        for i in idx:
            u = units[i]
            if u["flags"] & aacgpu.UNIT_CCE:
                continue
            for c in range(int(u["n_ch"])):
                ch = int(u["channel"]) + c
                data = spec[u["coef_offset"] + c].copy()
                for point in (0, 1):
                    for j in cces:
                        uj, rec = units[j], cce[units[j]["reserved1"]]
                        if rec["coupling_point"] != point:
                            continue
                        for t in range(int(rec["n_targets"])):
                            if rec["target"][t]["channel"] != ch:
                                continue
                            g = rec["gain"][rec["target"][t]["gain_list"]]
                            info = uj["ch"][0]
                            short = int(info["window_sequence"]) == 2
                            off = swb_short if short else swb_long
                            src = spec[uj["coef_offset"]]
                            base = 0
                            for grp in range(int(info["group_count"])):
                                for sfb in range(int(info["max_sfb"])):
                                    b = grp * int(info["max_sfb"]) + sfb
                                    if (meta[uj["meta_offset"]][b] >> 12) == 0:
                                        continue
                                    for w in range(int(info["group_len"][grp])):
                                        sl = slice(base + w * 128 + int(off[sfb]), base + w * 128 + int(off[sfb + 1]))
                                        data[sl] = (data[sl].astype(np.float64) + np.float64(g[b]) * src[sl].astype(np.float64)).astype(np.float32)
                                base += int(info["group_len"][grp]) * 128
                info = u["ch"][c]
                out = o.filterbank(int(info["window_sequence"]), int(info["window_shape"]), int(info["window_shape_prev"]), data, ovl[u["stream"], ch])
                for j in cces:
                    rec = cce[units[j]["reserved1"]]
                    if rec["coupling_point"] != 2:
                        continue
                    for t in range(int(rec["n_targets"])):
                        if rec["target"][t]["channel"] == ch:
                            g = np.float64(rec["gain"][rec["target"][t]["gain_list"]][0])
                            out = (out.astype(np.float64) + g * time_c[j].astype(np.float64)).astype(np.float32)
                pcm[int(u["pcm_offset"]) + ch:int(u["pcm_offset"]) + 1024 * C:C] = (out.astype(np.float64) / 32768.0).astype(np.float32)
    return pcm, ovl


CASES = [(("cpe", "sce"), (0, 1, 2)), (("cpe", "cpe", "cpe", "sce"), (2,)), (("cpe",), (0,)), (("sce", "cpe", "cpe", "sce"), (1, 2, 2))]


@pytest.mark.parametrize("layout,points", CASES)
def test_oracle_coupling_against_numpy_form(oracle, layout, points):
    S, T = 2, 4
    wl, units, q, meta, cce = workload(layout, points, S, T)
    H = wl["C"] + len(points)
    ov = np.zeros((S, H, 1024), np.float32)
    ref = oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce)
    want, want_ov = numpy_coupling(oracle, wl, units, q, meta, cce, S, H)
    assert np.array_equal(ref.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ov.view(np.uint32), want_ov.view(np.uint32))
    plain = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], np.zeros((S, H, 1024), np.float32))
    assert rel(ref, plain) > 1e-3                                 # the coupling does something


@pytest.mark.parametrize("layout,points", CASES)
def test_emulated_kernels_vs_oracle(oracle, layout, points):
    S, T = 2, 5
    wl, units, q, meta, cce = workload(layout, points, S, T)
    H = wl["C"] + len(points)
    ov = np.zeros((S, H, 1024), np.float32)
    ref = oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce)
    pool = np.zeros((S, H, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * H, np.uint8)
    got = emu_lib.Emu().decode(units, q, meta, wl["n_pcm"], pool, par, cce=cce)
    assert rel(got, ref) < RMS_REL
    assert np.abs(emu_lib.pool_current(pool, par) - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max()))


def test_fused_and_separate_independent_coupling_agree(oracle):
    """Independent coupling is applied where the target's PCM is formed (aacg_imdct_run_*_cpl: per-unit job lists, coupling
    elements' filterbank first) — except in plans with double-duty runs, which keep the separate pass over the interleaved
    PCM (aacg_couple_pcm).  Both routes on the same batches: the same samples bit for bit (one fused multiply-add per
    coupling on the finished sample either way, in the order of the frame's coupling elements), for first frames of chains
    (overlap from the state buffer), later frames (in place in the previous wave's slot), long and short windows."""
    for layout, points, T in [(("cpe", "cpe", "cpe", "sce"), (2, 2), 9), (("cpe",), (2,), 6), (("sce", "cpe"), (2, 0, 2), 5)]:
        S = 2
        wl, units, q, meta, cce = workload(layout, points, S, T)
        H = wl["C"] + len(points)
        outs = []
        for unfused in (False, True):
            pool = np.zeros((S, H, emu_lib.OV_BUFFERS, 1024), np.float32)
            outs.append(emu_lib.Emu().decode(units, q, meta, wl["n_pcm"], pool, np.zeros(S * H, np.uint8), cce=cce, unfused=unfused))
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
        ov = np.zeros((S, H, 1024), np.float32)
        assert rel(outs[0], oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce)) < RMS_REL


def test_planner_refuses_coupling_elements_without_the_mode(oracle):
    wl, units, q, meta, cce = workload(("cpe",), (0,))
    pool = np.zeros((2, 3, emu_lib.OV_BUFFERS, 1024), np.float32)
    with pytest.raises(RuntimeError, match="coupling channel element"):
        emu_lib.Emu().decode(units, q, meta, wl["n_pcm"], pool, np.zeros(6, np.uint8))          # no cce records: AACG_CCE_REFERENCE
    bad = cce.copy()
    bad["target"][0][0]["channel"] = 7
    with pytest.raises(RuntimeError, match="out of range"):
        emu_lib.Emu().decode(units, q, meta, wl["n_pcm"], pool, np.zeros(6, np.uint8), cce=bad)
    twice = cce.copy()                                            # the same target channel listed twice: two jobs of one launch would add to it
    twice["n_targets"][0] = 2
    twice["target"][0][1] = twice["target"][0][0]
    with pytest.raises(RuntimeError, match="twice"):
        emu_lib.Emu().decode(units, q, meta, wl["n_pcm"], pool, np.zeros(6, np.uint8), cce=twice)


@pytest.mark.gpu
@pytest.mark.parametrize("layout,points,T", [(("sce", "cpe", "cpe", "sce"), (2, 2), 15), (("cpe", "cpe", "cpe", "sce"), (2,), 20), (("cpe",), (2,), 31),
                                             (("sce", "cpe"), (2, 0, 2), 9)])
def test_gpu_fused_and_separate_independent_coupling_agree(oracle, layout, points, T):
    """on the GPU, the fused route (coupling applied in the targets' epilogues, aacg_imdct_run_quant_cpl)
    and the staged route (aacg_couple_pcm over the finished PCM; forced with aacg_debug_set_route) give the SAME BITS — 7 channels
    + coupling elements, two batches chained through the overlap state — and both stay at the oracle."""
    S = 12
    wl0 = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=21)
    H = wl0["C"] + len(points)
    outs, routes = [], []
    for unfused in (0, 1):
        eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=H, cce_mode=aacgpu.CCE_SPEC)
        eng.debug_set_route(unfused)
        got = []
        for batch in range(2):
            wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=21 + batch, frame_base=batch * T)
            units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=400 + batch)
            if batch == 0:
                plan = eng.plan(units, cce=cce)
                routes.append(eng.plan_kernels(plan))
                plan.destroy()
            got.append(eng.decode_batch(units, q, meta, wl["n_pcm"], cce=cce))
        outs.append(np.concatenate(got))
        eng.close()
    assert "_cpl" in routes[0] and "aacg_couple_pcm" in routes[1] and "_cpl" not in routes[1]
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    ov = np.zeros((S, H, 1024), np.float32)
    ref = []
    for batch in range(2):
        wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=21 + batch, frame_base=batch * T)
        units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=400 + batch)
        ref.append(oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce))
    assert rel(outs[0], np.concatenate(ref)) < RMS_REL


@pytest.mark.gpu
@pytest.mark.parametrize("layout,points,T,inp", [(("cpe", "sce"), (0, 1, 2), 5, "q"), (("cpe", "cpe", "cpe", "sce"), (2,), 20, "q"),
                                                (("cpe",), (0,), 40, "q"), (("sce", "cpe", "cpe", "sce"), (1, 2), 7, "q"), (("cpe", "cpe"), (2, 2, 0), 19, "q"),
                                                (("cpe", "sce"), (0, 2), 6, "spec")])
def test_gpu_coupling_vs_oracle(oracle, layout, points, T, inp):
    S = 6
    wl0 = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=11)
    H = wl0["C"] + len(points)
    kind = aacgpu.INPUT_QUANT_I16 if inp == "q" else aacgpu.INPUT_SPEC_F32
    eng = aacgpu.Engine(kind, max_streams=S, max_channels=H, cce_mode=aacgpu.CCE_SPEC)
    ov = np.zeros((S, H, 1024), np.float32)
    for batch in range(2):                                         # the second batch starts from the overlap state of the first
        wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=11 + batch, frame_base=batch * T)
        units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=300 + batch)
        if inp == "spec":                                          # FilterBank seam: the spectra every element would have dequantised to
            plain = units.copy()
            mono = plain[(plain["flags"] & aacgpu.UNIT_CCE) != 0].copy()
            mono["flags"], mono["n_out_ch"], mono["channel"] = 0, 1, 0
            mono["pcm_offset"] = np.arange(len(mono)) * 1024
            _, s1 = oracle.decode_batch(plain[(plain["flags"] & aacgpu.UNIT_CCE) == 0], q, meta, wl["n_pcm"], np.zeros((S, H, 1024), np.float32), want_spec=True)
            _, s2 = oracle.decode_batch(mono, q, meta, len(mono) * 1024, np.zeros((S, 1, 1024), np.float32), want_spec=True)
            coeffs, m = (s1 + s2).astype(np.float32), None
        else:
            coeffs, m = q, meta
        ref = oracle.decode_batch(units, coeffs, m, wl["n_pcm"], ov, cce=cce)
        got = eng.decode_batch(units, coeffs, m, wl["n_pcm"], cce=cce)
        assert np.isfinite(got).all() and rel(got, ref) < RMS_REL
        have = np.stack([[eng.get_overlap(s, c) for c in range(H)] for s in range(S)])
        assert np.abs(have - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max()))
    eng.close()


@pytest.mark.gpu
def test_gpu_coupling_with_tns(oracle):
    """Coupling before and after the TNS filter of the target (decoder.js:258-266): AACG_CCE_SPEC + AACG_TNS_SPEC."""
    S, T, layout, points = 4, 6, ("cpe", "sce"), (0, 1)
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=21)
    tu, tns = aacgpu_workload.add_tns_config3(wl, seed=5)
    wl["units"] = tu
    units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=7)
    H = wl["C"] + len(points)
    ov = np.zeros((S, H, 1024), np.float32)
    ref = oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, tns=tns, cce=cce)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=H, tns_mode=aacgpu.TNS_SPEC, cce_mode=aacgpu.CCE_SPEC)
    got = eng.decode_batch(units, q, meta, wl["n_pcm"], tns=tns, cce=cce)
    assert rel(got, ref) < 1e-5
    only_tns = oracle.decode_batch(tu, wl["q"], wl["meta"], wl["n_pcm"], np.zeros((S, H, 1024), np.float32), tns=tns)
    assert rel(ref, only_tns) > 1e-3
    eng.close()


@pytest.mark.gpu
def test_gpu_coupling_plan_and_errors(oracle):
    """The device-resident path (aacg_plan_create_ex) relaunched, and the refusals."""
    import torch
    S, T, layout, points = 8, 16, ("cpe", "cpe", "cpe", "sce"), (2,)
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=31)
    units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=9)
    H = wl["C"] + 1
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=H, cce_mode=aacgpu.CCE_SPEC)
    plan = eng.plan(units, cce=cce)
    d_q, d_meta = torch.from_numpy(q).cuda(), torch.from_numpy(meta.view(np.int16)).cuda()
    d_pcm = torch.zeros(wl["n_pcm"], dtype=torch.float32, device="cuda")
    ov = np.zeros((S, H, 1024), np.float32)
    for launch in range(2):
        ref = oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce)
        eng.decode_device(plan, d_q.data_ptr(), d_meta.data_ptr(), d_pcm.data_ptr())
        eng.synchronize()
        assert rel(d_pcm.cpu().numpy(), ref) < RMS_REL
    plan.destroy()
    eng.close()
    plain = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=H)               # AACG_CCE_REFERENCE
    with pytest.raises(aacgpu.AacgError) as ei:
        plain.decode_batch(units, q, meta, wl["n_pcm"], cce=cce)
    assert ei.value.code == -5 and "coupling" in str(ei.value)
    plain.close()
    with pytest.raises(aacgpu.AacgError):
        aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=1, max_channels=3, cce_mode=aacgpu.CCE_SPEC, output_kind=aacgpu.OUTPUT_I16)


# ---- through the JavaScript host: front end keeps the coupling elements, GpuAACDecoder hands them to the engine ----
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="node / node_api.h not present")


def js_records(tmp, mode):
    out = str(tmp)
    subprocess.run(["make", "-C", os.path.join(ROOT, "aac.js_amd", "napi")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_coupling.js"), out, mode], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "coupling %s tests ok" % mode in r.stdout, r.stdout + r.stderr
    f = lambda ext, dt: np.fromfile(os.path.join(out, "coupling" + ext), dt)
    info = json.load(open(os.path.join(out, "coupling.json")))
    return info, f(".units", np.uint8).view(orc.UNIT_DTYPE).ravel(), f(".q", np.int16), f(".meta", np.uint16), f(".cce", np.uint8).view(orc.CCE_DTYPE).ravel(), out


@needs_node
def test_js_host_coupling_records(oracle, tmp_path):
    """tests/js/test_coupling.js checks what the front end parsed (targets, every gain) and how the decoder resolved it; the
    records it hands to the engine then decode the same in the oracle and in the emulated kernels, and differently from the
    same stream with the coupling elements dropped."""
    info, units, q, meta, cce, _ = js_records(tmp_path, "cpu")
    C, H, n = info["channels"], info["channels"] + info["hidden"], info["frames"]
    ov = np.zeros((1, H, 1024), np.float32)
    ref = oracle.decode_batch(units, q, meta, n * 1024 * C, ov, cce=cce)
    pool = np.zeros((1, H, emu_lib.OV_BUFFERS, 1024), np.float32)
    got = emu_lib.Emu().decode(units, q, meta, n * 1024 * C, pool, np.zeros(H, np.uint8), cce=cce)
    assert rel(got, ref) < RMS_REL
    dropped = oracle.decode_batch(units[(units["flags"] & aacgpu.UNIT_CCE) == 0], q, meta, n * 1024 * C, np.zeros((1, H, 1024), np.float32))
    assert rel(ref, dropped) > 1e-3


@pytest.mark.gpu
@needs_node
def test_js_host_coupling_gpu(oracle, tmp_path):
    """bytes -> FrontEnd (coupling kept) -> GpuAACDecoder({ cceMode: CCE_SPEC }).readChunk() on the real engine == the oracle on
    the records the same decoder produced."""
    info, units, q, meta, cce, out = js_records(tmp_path, "gpu")
    C, H, n = info["channels"], info["channels"] + info["hidden"], info["frames"]
    ref = oracle.decode_batch(units, q, meta, n * 1024 * C, np.zeros((1, H, 1024), np.float32), cce=cce)
    pcm = np.fromfile(os.path.join(out, "coupling.pcm"), np.float32)
    assert pcm.size == ref.size and rel(pcm, ref) < RMS_REL


@pytest.mark.gpu
@pytest.mark.parametrize("layout,points,T,S", [(("cpe", "cpe", "cpe", "sce"), (2,), 16, 64), (("sce", "cpe"), (2, 2), 9, 40)])
def test_gpu_coupling_through_the_pipeline_equals_launch_behind_launch(oracle, layout, points, T, S):
    """Batches with fused independent coupling through aacg_decode_pipelined (no rendezvous build: launch behind launch on the
    pipeline's first stream): 24 launches of one plan that way against the same launches through aacg_decode_device — the same
    bits, PCM and overlap state; the first launch against the oracle.  (Round 5 also ran the coupling elements' pass of launch
    n + 1 beside the main launch n, on a stream and a side buffer of its own: the same bits, 90.8 us against 87.3 — the two
    kernels contend for the chip and the two event waits per step cost more than the pass's ramp and drain; not kept.)"""
    import torch
    n = 24
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, seed=31)
    H = wl["C"] + len(points)
    units, q, meta, cce = aacgpu_workload.add_cce(wl, points=points, seed=500)
    ins = [torch.from_numpy(np.ascontiguousarray(np.roll(q, 53 * j, axis=0))).cuda() for j in range(n)]
    dm = torch.from_numpy(meta.view(np.int16)).cuda()
    results = []
    for pipelined in (False, True):
        eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=H, cce_mode=aacgpu.CCE_SPEC)
        plan = eng.plan(units, cce=cce)
        assert "_cpl" in eng.plan_kernels(plan, pipelined=pipelined)
        outs = [torch.full((wl["n_pcm"],), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        for j in range(n):
            if pipelined:
                eng.decode_pipelined(plan, ins[j].data_ptr(), dm.data_ptr(), outs[j].data_ptr())
            else:
                eng.decode_device(plan, ins[j].data_ptr(), dm.data_ptr(), outs[j].data_ptr(), 0)
        eng.synchronize()
        torch.cuda.synchronize()
        state = np.stack([[eng.get_overlap(s, c) for c in range(H)] for s in range(S)])
        results.append(([o.cpu().numpy() for o in outs], state))
        plan.destroy()
        eng.close()
    (serial, s_state), (piped, p_state) = results
    ov = np.zeros((S, H, 1024), np.float32)
    assert rel(serial[0], oracle.decode_batch(units, q, meta, wl["n_pcm"], ov, cce=cce)) < RMS_REL
    for j in range(n):
        assert not np.isnan(piped[j]).any(), j
        assert np.array_equal(piped[j].view(np.uint32), serial[j].view(np.uint32)), "launch %d differs from the serialised route" % j
    assert np.array_equal(p_state.view(np.uint32), s_state.view(np.uint32))
