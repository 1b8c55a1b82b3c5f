"""The one-channel-per-wave run kernels (aac.js_amd/csrc/aacg_kernels8.h: 8 waves per SIMD, two workgroups per CU) in the lane
emulator: against the oracle, the planner's run table for them, and the properties the run-to-run rendezvous must have —
whichever side of a hand-over arrives first, and however a chain is cut into batches, the same BITS come out."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu_workload as W  # noqa: E402
import emu_lib  # noqa: E402

RMS_TOL = 1e-5


@pytest.fixture(scope="module")
def emu():
    return emu_lib.Emu()


def _rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def _decode(emu, wl, coeffs, meta, S, C, run8, pool=None, par=None):
    pool = np.zeros((S, C, 2, 1024), np.float32) if pool is None else pool
    par = np.zeros(S * C, np.uint8) if par is None else par
    return emu.decode(wl["units"], coeffs, meta, wl["n_pcm"], pool, par, run8=run8), pool, par


@pytest.mark.parametrize("layout,S,T,seam", [(("cpe",), 2, 20, "q"), (("cpe",), 2, 20, "f"), (("sce",), 1, 35, "q"), (("cpe", "cpe", "cpe", "sce"), 1, 9, "q"),
                                             (("sce", "cpe"), 1, 17, "f")])
def test_run8_matches_the_oracle_whichever_side_of_a_rendezvous_comes_first(emu, oracle, layout, S, T, seam):
    """Chains longer than a run (8 frames of a pair, 16 of a single channel) hand their tail over through a rendezvous cell.  The
    emulator runs the workgroups one after the other: in block order the publishing side is always first, in reverse order the
    consuming side — both orders must give the oracle's PCM and overlap state, and the same bits as each other."""
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=31)
    C = wl["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    coeffs, meta = (wl["q"], wl["meta"]) if seam == "q" else (spec.astype(np.float32), None)
    got = [_decode(emu, wl, coeffs, meta, S, C, r8) for r8 in (1, 2)]
    assert _rms(got[0][0], ref) < RMS_TOL and not np.isnan(got[0][0]).any()
    assert np.array_equal(got[0][0].view(np.uint32), got[1][0].view(np.uint32))
    state = [emu_lib.pool_current(g[1], g[2]) for g in got]
    assert np.array_equal(state[0].view(np.uint32), state[1].view(np.uint32))
    assert np.abs(state[0] - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max()))


def test_run8_is_bit_identical_however_the_batch_is_cut(emu):
    """The overlap-add is tail + head with both terms rounded products, never a fused multiply-add: one batch of 21 frames, or 8 + 13,
    or 1 + 20 through the overlap state in between — the same bits."""
    S, T = 2, 21
    whole = W.make_batch(n_streams=S, n_frames=T, mix=True, intensity=True, seed=77)
    ref_pcm, _, _ = _decode(emu, whole, whole["q"], whole["meta"], S, 2, 1)
    ref_pcm = ref_pcm.reshape(S, T, 2048)
    per_frame = len(whole["units"]) // (S * T)
    for cut in (8, 1):
        pool, par = np.zeros((S, 2, 2, 1024), np.float32), np.zeros(S * 2, np.uint8)
        parts = []
        for lo, hi in ((0, cut), (cut, T)):
            keep = np.zeros(len(whole["units"]), bool)
            for s in range(S):
                keep[(s * T + lo) * per_frame:(s * T + hi) * per_frame] = True
            units = whole["units"][keep].copy()
            for s in range(S):
                sel = units["stream"] == s
                units["pcm_offset"][sel] = units["pcm_offset"][sel] - units["pcm_offset"][sel].min() + s * (hi - lo) * 2048
            pcm = emu.decode(units, whole["q"], whole["meta"], S * (hi - lo) * 2048, pool, par, run8=1)
            parts.append(pcm.reshape(S, hi - lo, 2048))
        got = np.concatenate(parts, axis=1)
        assert np.array_equal(got.view(np.uint32), ref_pcm.view(np.uint32)), cut


def test_planner_cuts_chains_for_run8(emu):
    """8 frames of a pair / 16 of a single channel per run; consecutive runs of a chain linked by one rendezvous cell each, the
    successor's first unit recorded (the run that arrives second finishes that frame)."""
    wl = W.make_batch(n_streams=3, n_frames=20, layout=("cpe", "sce"), seed=3)
    runs, n_links = emu.plan8(wl["units"], 3, 3)
    pair = runs[runs["n_ch"] == 2]
    mono = runs[runs["n_ch"] == 1]
    assert sorted(pair["n_units"].tolist()) == [4] * 3 + [8] * 6 and sorted(mono["n_units"].tolist()) == [4] * 3 + [16] * 3
    assert n_links == 3 * 2 + 3 * 1
    outs = runs["link_out"][runs["link_out"] >= 0]
    ins = runs["link_in"][runs["link_in"] >= 0]
    assert sorted(outs.tolist()) == sorted(ins.tolist()) == list(range(n_links))
    for r in runs:
        assert (r["link_out"] >= 0) == (r["succ_unit"] >= 0)
        if r["link_out"] >= 0:
            nxt = runs[runs["link_in"] == r["link_out"]][0]
            assert nxt["unit"][0] == r["succ_unit"] and nxt["n_ch"] == r["n_ch"]


def test_run8_and_the_wide_kernels_agree_to_rounding(emu):
    """The two kernel families round differently (the pair kernels may fuse the window product into the overlap-add); both stay at
    the oracle, a few ulp from each other."""
    wl = W.make_batch(n_streams=2, n_frames=12, mix=True, seed=5)
    a, _, _ = _decode(emu, wl, wl["q"], wl["meta"], 2, 2, 1)
    b, _, _ = _decode(emu, wl, wl["q"], wl["meta"], 2, 2, 0)
    assert _rms(a, b) < 2e-7 and np.abs(a - b).max() < 4e-6
