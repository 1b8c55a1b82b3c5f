"""Chains longer than a run on the 16-wave kernels WITHOUT a recomputed frame (imdct_run_body<..., RV>, aacg_engine_rv.hip): the runs of
a chain meet in a rendezvous cell in global memory — whichever side arrives first publishes (the windowed tail, or the windowed
first half of the next frame), the second finishes the frame.  In the lane emulator the workgroups run one after the other, in
block order (the publishing run first) or reversed (the consuming run first): both orders, and the old route that recomputes a
frame per later run (_dd kernels), must give the same BITS — PCM and overlap state — and the oracle's values."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu_workload as W  # noqa: E402
import emu_lib  # noqa: E402


@pytest.mark.parametrize("layout,S,T,seam", [(("cpe",), 2, 37, "q"), (("cpe",), 1, 33, "f"), (("sce",), 1, 50, "q"), (("cpe", "cpe", "cpe", "sce"), 1, 20, "q"),
                                             (("sce", "cpe"), 1, 17, "f")])
def test_rendezvous_route_equals_the_recompute_route_bit_for_bit(oracle, layout, S, T, seam):
    emu = emu_lib.Emu()
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=61)
    C = wl["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    coeffs, meta = (wl["q"], wl["meta"]) if seam == "q" else (spec.astype(np.float32), None)
    got = []
    for rv in (1, 2, 0):
        pool, par = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32), np.zeros(S * C, np.uint8)
        pcm = emu.decode(wl["units"], coeffs, meta, wl["n_pcm"], pool, par, rv=rv)
        got.append((pcm, emu_lib.pool_current(pool, par)))
    d = got[0][0].astype(np.float64) - ref
    assert float(np.sqrt(np.mean(d * d))) < 1e-5 and not np.isnan(got[0][0]).any()
    for other in got[1:]:
        assert np.array_equal(got[0][0].view(np.uint32), other[0].view(np.uint32))
        assert np.array_equal(got[0][1].view(np.uint32), other[1].view(np.uint32))
    assert np.abs(got[0][1] - ov).max() <= 1e-5 * max(1.0, float(np.abs(ov).max()))
