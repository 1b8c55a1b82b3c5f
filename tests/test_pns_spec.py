"""AACG_PNS_SPEC: NOISE_BT bands as ics.js:228-243 was meant to fill them.

aac.js's own generator degenerates (ics.js:234: `randomState * (1664525 + 1013904223)`) and its output turns to
NaN after 11 draws, so there is no reference output to pin to: PARITY UNPINNED BY THE REFERENCE for this mode.
The oracle's restatement (orc_dequant_pns) is checked against an independent numpy restatement below; the
kernels against the oracle (they take a band-window's energy as a difference of running sums, the oracle sums
sequentially: the scale differs in the last bits of a double, the float32 result by at most an ulp).
The default mode refuses such batches, like the reference cannot decode them.
"""
import numpy as np
import pytest

import aacgpu_workload as W
import emu_lib
import orc

REL_TOL = 5e-6


def numpy_pns(info, band_words, sample_rate_tables=None):
    """The noise part of one channel's spectrum (zeros elsewhere), float32, ICStream.data order."""
    short = int(info["window_sequence"]) == 2
    swb = W.SWB_SHORT_48 if short else W.SWB_LONG_48
    out = np.zeros(1024, np.float32)
    state = np.uint32(0x1F2E3D4C)
    group_off, idx = 0, 0
    for g in range(int(info["group_count"])):
        glen = int(info["group_len"][g])
        for sfb in range(int(info["max_sfb"])):
            word = int(band_words[idx]); idx += 1
            if word >> 12 != 13:
                continue
            sf = np.float32(2.0 ** (((word & 0x1ff) - 200) / 4.0))
            if word & 0x200:
                sf = -sf
            lo, width = int(swb[sfb]), int(swb[sfb + 1] - swb[sfb])
            for w in range(glen):
                vals = np.zeros(width, np.float32)
                for k in range(width):
                    state = np.uint32((int(state) * 1664525 + 1013904223) & 0xFFFFFFFF)
                    vals[k] = np.float32(np.int32(state))
                energy = float(np.sum(vals.astype(np.float64) ** 2))
                scale = float(sf) / np.sqrt(energy)
                out[group_off + w * 128 + lo: group_off + w * 128 + lo + width] = (vals.astype(np.float64) * scale).astype(np.float32)
        group_off += glen * 128
    return out


def _noise_batch(seed, **kw):
    wl = W.random_batch(seed, **kw) if kw.get("max_frames") else W.make_batch(seed=seed, **kw)
    units, meta = W.add_pns(wl, seed=seed)
    return wl, units, meta


def test_oracle_pns_matches_numpy(oracle):
    wl, units, meta = _noise_batch(31, n_streams=2, max_frames=6)
    S, C = wl["n_streams"], wl["max_channels"]
    ov = np.zeros((S, C, 1024), np.float32)
    _, spec = oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, want_spec=True, pns=True)
    checked = 0
    for u in units:
        if not (int(u["flags"]) & 4) or (int(u["n_ch"]) == 2 and (int(u["flags"]) & 3) == 3):
            continue                                    # MS mixes the channels of masked CPEs: check the plain ones
        for c in range(int(u["n_ch"])):
            words = meta[int(u["meta_offset"]) + c]
            want = numpy_pns(u["ch"][c], words)
            got = spec[int(u["coef_offset"]) + c]
            mask = want != 0
            if not mask.any():
                continue
            if int(u["n_ch"]) == 2 and c == 1 and np.any((words >> 12) >= 14):
                continue                                # intensity bands rewrite the right channel
            assert np.max(np.abs(got[mask] - want[mask])) <= 2e-6 * np.max(np.abs(want[mask])), (int(u["coef_offset"]), c)
            checked += 1
    assert checked >= 4
    # the sequence itself: the intended LCG, restarted per channel — the first noise band of a channel starts at draw 1
    with pytest.raises(RuntimeError):
        oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov)            # REFERENCE mode refuses


@pytest.fixture(scope="module")
def emu():
    return emu_lib.Emu()


def _rel(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b * b)), 1e-30))


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_emulated_kernels_pns_vs_oracle(emu, oracle, seed):
    wl, units, meta = _noise_batch(seed, n_streams=2, max_frames=5)
    S, C = wl["n_streams"], wl["max_channels"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, pns=True)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    pcm = emu.decode(units, wl["q"], meta, wl["n_pcm"], pool, par, pns=True)
    assert _rel(pcm, ref) < REL_TOL
    assert _rel(emu_lib.pool_current(pool, par), ov) < REL_TOL
    with pytest.raises(RuntimeError, match="rc=-5"):
        emu.decode(units, wl["q"], meta, wl["n_pcm"], pool, par)               # REFERENCE mode refuses


def test_emulated_pns_long_chain_and_grouped_shorts(emu, oracle):
    wl = W.make_batch(n_streams=1, n_frames=34, mix=True, intensity=True, seed=77)      # 16 + 16 + 2 frames
    units, meta = W.add_pns(wl, seed=5, p_unit=1.0, p_band=0.4)
    ov = np.zeros((1, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, pns=True)
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    pcm = emu.decode(units, wl["q"], meta, wl["n_pcm"], pool, par, pns=True)
    assert _rel(pcm, ref) < REL_TOL


# ---- the HIP path ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [301, 302, 303, 304])
def test_gpu_pns_vs_oracle(oracle, seed):
    import aacgpu
    wl, units, meta = _noise_batch(seed, n_streams=4, max_frames=24)
    S, C = wl["n_streams"], wl["max_channels"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, pns=True)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C, pns_mode=aacgpu.PNS_SPEC)
    pcm = eng.decode_batch(units, wl["q"], meta, wl["n_pcm"])
    assert _rel(pcm, ref) < REL_TOL
    got = np.stack([[eng.get_overlap(s, c) for c in range(C)] for s in range(S)])
    assert _rel(got, ov) < REL_TOL
    # batches without noise bands take the ordinary kernel on the same engine
    ov2 = ov.copy()
    ref2 = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov2)
    pcm2 = eng.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"])
    assert _rel(pcm2, ref2) < REL_TOL
    eng.close()
    # the default engine refuses: by the unit flag at plan time, by the band type on the host-buffer path
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C)
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.decode_batch(units, wl["q"], meta, wl["n_pcm"])
    assert ei.value.code == -5
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.decode_batch(wl["units"], wl["q"], meta, wl["n_pcm"])             # noise bands without the flag
    assert ei.value.code == -5
    eng.close()


@pytest.mark.gpu
def test_gpu_pns_device_path_with_tns(oracle):
    """The plan / device-pointer path, together with TNS SPEC (the PNS stage feeds the f32 TNS kernel)."""
    import aacgpu
    import torch
    S, T = 3, 20
    wl = W.make_batch(n_streams=S, n_frames=T, mix=True, intensity=True, seed=88)
    units, meta = W.add_pns(wl, seed=8)
    units, tns = W.add_tns(dict(units=units), seed=9)
    ov = np.zeros((S, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], meta, wl["n_pcm"], ov, tns=tns, pns=True)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, 2, tns_mode=aacgpu.TNS_SPEC, pns_mode=aacgpu.PNS_SPEC)
    plan = eng.plan(units, tns=tns)
    dq = torch.from_numpy(wl["q"]).cuda()
    dm = torch.from_numpy(meta.view(np.int16)).cuda()
    dp = torch.zeros(wl["n_pcm"], dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.decode_device(plan, dq.data_ptr(), dm.data_ptr(), dp.data_ptr(), 0)
    eng.synchronize()
    assert _rel(dp.cpu().numpy(), ref) < 2e-5
    plan.destroy()
    eng.close()
