"""Consecutive launches of one plan OVERLAPPED (aacg_decode_pipelined; imdct_run_body<..., RV> with cross-launch cells): the last
frame of a chain in launch k and its first frame in launch k + 1 meet in a rendezvous cell — whichever side arrives first
publishes (the windowed tail = the new overlap state, or the windowed first half plus where the finished samples go), the
second finishes the frame; nobody waits (reference: the hand-over of filter_bank.js:105-118 through `overlaps`, :38-41).

The lane emulator runs one workgroup at a time, in every kind of order the engine's ordering rules allow (launch n behind
launch n - 2 by its stream, every other launch of a stream behind launch n - 3 by an event): launch after launch, the LATER
launch of every pair first, and random interleavings of everything those rules let be in flight.  All of them must give
the BITS of the serialised route (one aacg_decode_device after the other: the plain kernels) — PCM of every launch and the
overlap state at the end — and the oracle's values."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu_workload as W  # noqa: E402
import emu_lib  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    return emu_lib.Emu()


def _cells(S, C):
    """aacg_xl_cell records (4 x u64 each) with rubbish in them — a state word counts only with the right epoch — and the head pool"""
    cells = np.full((S, C, emu_lib.OV_BUFFERS, 4), 0x5a5a5a5a5a5a5a5a, np.uint64)
    heads = np.full((S, C, emu_lib.OV_BUFFERS, 1024), np.nan, np.float32)
    return cells, heads


def _batches(S, T, layout, n, seam, oracle, seed):
    """n consecutive batches of the same streams (same plan structure: the unit records of batch 0 serve all of them, like a
    relaunched plan), the oracle's PCM for each and its final overlap state"""
    base = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=False, seed=seed)
    C = base["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    rng = np.random.default_rng(seed)
    coeffs, metas, refs = [], [], []
    for j in range(n):
        q = base["q"] if j == 0 else (np.roll(base["q"], 37 * j, axis=0) * rng.choice([-1, 1])).astype(np.int16)
        ref, spec = oracle.decode_batch(base["units"], q, base["meta"], base["n_pcm"], ov, want_spec=True)
        refs.append(ref)
        coeffs.append(q if seam == "q" else spec.astype(np.float32))
        metas.append(base["meta"])
    return base, C, coeffs, (metas if seam == "q" else None), refs, ov


@pytest.mark.parametrize("layout,S,T,n,seam", [(("cpe",), 3, 16, 7, "q"), (("cpe",), 2, 7, 6, "f"), (("sce",), 2, 16, 4, "q"),
                                                (("cpe",), 1, 37, 4, "q"), (("cpe", "cpe", "cpe", "sce"), 1, 5, 6, "q"), (("sce", "cpe"), 1, 18, 3, "f"),
                                                (("cpe",), 2, 3, 2 * emu_lib.OV_BUFFERS + 3, "q")])       # the rotating buffers and cells come round twice
def test_overlapped_launches_equal_the_serialised_route_bit_for_bit(emu, oracle, layout, S, T, n, seam):
    base, C, coeffs, metas, refs, ov = _batches(S, T, layout, n, seam, oracle, 71)
    # the serialised route: one launch after the other, each from the complete state the one before left
    pool, par = emu_lib.new_pool(S, C)
    serial = [emu.decode(base["units"], coeffs[j], metas[j] if metas else None, base["n_pcm"], pool, par) for j in range(n)]
    serial_state = emu_lib.pool_current(pool, par)
    for j in range(n):
        d = serial[j].astype(np.float64) - refs[j]
        assert float(np.sqrt(np.mean(d * d))) < 1e-5 and not np.isnan(serial[j]).any()
    assert np.abs(serial_state - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max()))
    for order, streams in ((0, 0), (1, 0), (2, 0), (3, 0), (11, 0), (1, 2), (5, 2)):        # streams 0: the engine's choice for the plan (three here)
        pool, par = emu_lib.new_pool(S, C)
        cells, heads = _cells(S, C)
        got, _ = emu.decode_pipelined(base["units"], coeffs, metas, base["n_pcm"], pool, par, cells, heads, order=order, streams=streams)
        for j in range(n):
            assert np.array_equal(got[j].view(np.uint32), serial[j].view(np.uint32)), (order, j)
        assert np.array_equal(emu_lib.pool_current(pool, par).view(np.uint32), serial_state.view(np.uint32)), order


def test_a_sequence_continues_across_calls_and_into_the_serial_route(emu, oracle):
    """launches 0-2 pipelined, launch 3 as a continuation whose predecessor is 'still in flight' (epoch_in = its epoch), then a
    serial launch from the state the pipeline left: every step the serial route's bits"""
    S, T, n = 2, 16, 5
    base, C, coeffs, metas, refs, _ = _batches(S, T, ("cpe",), n, "q", oracle, 5)
    pool, par = emu_lib.new_pool(S, C)
    serial = [emu.decode(base["units"], coeffs[j], metas[j], base["n_pcm"], pool, par) for j in range(n)]
    serial_state = emu_lib.pool_current(pool, par)
    pool, par = emu_lib.new_pool(S, C)
    cells, heads = _cells(S, C)
    a, epoch = emu.decode_pipelined(base["units"], coeffs[:3], metas[:3], base["n_pcm"], pool, par, cells, heads, order=1)
    b, _ = emu.decode_pipelined(base["units"], coeffs[3:4], metas[3:4], base["n_pcm"], pool, par, cells, heads, order=0, epoch_in=epoch)
    c = emu.decode(base["units"], coeffs[4], metas[4], base["n_pcm"], pool, par)
    for j, g in enumerate(a + b + [c]):
        assert np.array_equal(g.view(np.uint32), serial[j].view(np.uint32)), j
    assert np.array_equal(emu_lib.pool_current(pool, par).view(np.uint32), serial_state.view(np.uint32))


def test_window_switching_across_the_launch_boundary(emu, oracle):
    """EIGHT_SHORT / START / STOP frames at the ends of the chains (the mix of BASELINE config 3): the cross-launch hand-over
    carries the zero stretch of a short frame's first half and every window shape"""
    S, T, n = 2, 8, 4
    base = W.make_batch(n_streams=S, n_frames=T, layout=("cpe",), mix=True, intensity=True, seed=91)
    C = base["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    coeffs = [np.roll(base["q"], 5 * j, axis=0) for j in range(n)]
    refs = [oracle.decode_batch(base["units"], coeffs[j], base["meta"], base["n_pcm"], ov) for j in range(n)]
    pool, par = emu_lib.new_pool(S, C)
    serial = [emu.decode(base["units"], coeffs[j], base["meta"], base["n_pcm"], pool, par) for j in range(n)]
    for j in range(n):
        d = serial[j].astype(np.float64) - refs[j]
        assert float(np.sqrt(np.mean(d * d))) < 1e-5
    for order in (1, 7):
        pool, par = emu_lib.new_pool(S, C)
        cells, heads = _cells(S, C)
        got, _ = emu.decode_pipelined(base["units"], coeffs, [base["meta"]] * n, base["n_pcm"], pool, par, cells, heads, order=order)
        for j in range(n):
            assert np.array_equal(got[j].view(np.uint32), serial[j].view(np.uint32)), (order, j)
