"""The kernels' source (aac.js_amd/csrc/aacg_kernels.h) executed lane-by-lane on CPU threads
(tests/emu) against the reference's golden vectors and the oracle.  This is the CPU-side check of
the wavefront choreography — lane maps, radix-8 stages, LDS transposes, mirror shuffles, window/OLA,
run hand-off, planner — since the build container has no GPU.  The same comparisons run against
the real HIP build in test_gpu_parity.py.
"""
import numpy as np
import pytest

import emu_lib
import orc

RMS_TOL = 1e-5          # internal gate; BASELINE.json's bound is 1e-4 RMS on [-1,1) PCM


@pytest.fixture(scope="module")
def emu():
    return emu_lib.Emu()


REL_TOL = 5e-6          # relative to the signal RMS; the reference's own IMDCT is 1.3e-6 from exact


def rms(a, b):
    """RMS error on the [-1,1) PCM scale; also gates the error relative to the signal level."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = a - b
    err = float(np.sqrt(np.mean(d * d)))
    sig = float(np.sqrt(np.mean(b * b)))
    assert err <= REL_TOL * sig + 1e-9, "relative RMS error %.3e (signal rms %.3e)" % (err / max(sig, 1e-30), sig)
    return err


def test_tables_match_reference(emu, golden):
    sl, kl, ss, ks = emu.windows()
    for got, name in ((sl, "tables.sine_long"), (kl, "tables.kbd_long"), (ss, "tables.sine_short"), (ks, "tables.kbd_short")):
        assert np.array_equal(got.view(np.uint32), golden[name].view(np.uint32)), name
    iq, sf = emu.iq_sf()
    assert np.array_equal(iq[:8191].view(np.uint32), golden["tables.iq"].view(np.uint32))
    assert np.isnan(iq[8191])
    assert np.array_equal(sf.view(np.uint32), golden["tables.sf"].view(np.uint32))


SCENARIOS = ["scn_stereo", "scn_split", "scn_7ch", "scn_mono"]


@pytest.mark.parametrize("name", SCENARIOS)
@pytest.mark.parametrize("inp", ["q", "spec"])
def test_scenarios_vs_reference(emu, golden, name, inp):
    units = golden[name + ".units"].view(orc.UNIT_DTYPE).ravel()
    ref = golden[name + ".pcm"]
    C = ref.shape[2]
    pool = np.zeros((1, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(C, np.uint8)
    pcm = emu.decode(units, golden[name + "." + inp], golden[name + ".meta"] if inp == "q" else None, ref.size, pool, par)
    assert not np.isnan(pcm).any()
    assert rms(pcm, ref) < RMS_TOL
    ov = emu_lib.pool_current(pool, par)[0]
    scale = max(1.0, float(np.abs(golden[name + ".overlap"]).max()))
    assert np.abs(ov - golden[name + ".overlap"]).max() / scale < 1e-5


@pytest.mark.parametrize("name", SCENARIOS)
def test_spectral_stage_bit_exact(emu, golden, name):
    """dequant + MS + IS are integer lookups and single fp32 operations: bit-exact (SURVEY §8a rows 3,6,7)."""
    units = golden[name + ".units"].view(orc.UNIT_DTYPE).ravel()
    spec = emu.spectral(units, golden[name + ".q"], golden[name + ".meta"])
    assert np.array_equal(spec.view(np.uint32), golden[name + ".spec"].view(np.uint32))


def test_cfg1(emu, golden):
    units = np.zeros(1, orc.UNIT_DTYPE)
    units["n_out_ch"] = 1
    units["n_ch"] = 1
    units["ch"]["max_sfb"][0, 0] = 49
    units["ch"]["group_count"][0, 0] = 1
    units["ch"]["group_len"][0, 0, 0] = 1
    pool = np.zeros((1, 1, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(1, np.uint8)
    pcm = emu.decode(units, golden["cfg1.spec"], None, 1024, pool, par)
    assert rms(pcm, golden["cfg1.pcm"]) < RMS_TOL


def test_batches_chain_like_one_long_batch(emu, golden, oracle):
    """Splitting a stream into consecutive batches (state through the double-buffered overlap pool)
    gives the same PCM as one batch; runs longer than AACG_RUN_FRAMES recompute their predecessor."""
    name = "scn_stereo"
    units = golden[name + ".units"].view(orc.UNIT_DTYPE).ravel().copy()
    ref = golden[name + ".pcm"]
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    got = np.empty(ref.size, np.float32)
    for lo, hi in ((0, 5), (5, 6), (6, 18)):                    # 5 + 1 + 12 frames
        u = units[lo:hi].copy()
        base = int(u["pcm_offset"][0])
        u["pcm_offset"] -= base
        cb = int(u["coef_offset"][0])
        u["coef_offset"] -= cb
        u["meta_offset"] -= cb
        pcm = emu.decode(u, golden[name + ".q"][cb:cb + 2 * (hi - lo)], golden[name + ".meta"][cb:cb + 2 * (hi - lo)],
                         (hi - lo) * 2048, pool, par)
        got[base:base + pcm.size] = pcm
    assert rms(got, ref) < RMS_TOL


def _workload(**kw):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    return aacgpu_workload.make_batch(**kw)


@pytest.mark.parametrize("mix,layout,intensity", [(False, ("cpe",), False), (True, ("cpe",), True),
                                                  (True, ("cpe", "cpe", "cpe", "sce"), False), (True, ("sce",), False)])
def test_synthetic_multistream_vs_oracle(emu, oracle, mix, layout, intensity):
    """BASELINE configs 2/3/5 at reduced size: several streams x 19 frames (two runs per chain: 16 + 3)."""
    S, T = 3, 19
    wl = _workload(n_streams=S, n_frames=T, layout=layout, mix=mix, intensity=intensity, seed=1234)
    C = wl["C"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref, spec_ref = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    pcm = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    assert rms(pcm, ref) < RMS_TOL
    assert np.abs(emu_lib.pool_current(pool, par) - ov).max() < 1e-5 * max(1.0, np.abs(ov).max())
    spec = emu.spectral(wl["units"], wl["q"], wl["meta"])
    assert np.array_equal(spec.view(np.uint32), spec_ref.view(np.uint32))
    # filterbank seam on the same data
    pool[:] = 0
    par[:] = 0
    pcm2 = emu.decode(wl["units"], spec_ref, None, wl["n_pcm"], pool, par)
    assert rms(pcm2, ref) < RMS_TOL


def test_planner_runs(emu):
    T = 37                                                     # 16 + 15 + 6 per chain
    wl = _workload(n_streams=5, n_frames=T)
    n, runs, info = emu.plan(wl["units"], 5, 2)
    assert n == 5 * 3 and info[1] == 5 and info[0] == 0
    seen = np.zeros(len(wl["units"]), int)
    for r in runs:
        ids = r["unit"][:r["n_units"]]
        assert (np.diff(ids) == 1).all()                       # consecutive frames of one stream
        seen[ids] += 1
        first = ids[0] % T == 0
        assert (r["pred_unit"] == -1) == first and (first or r["pred_unit"] == ids[0] - 1)
        assert r["n_units"] <= (16 if first else 15)           # wave 0 of a later run recomputes the predecessor
        assert r["is_last"] == (ids[-1] % T == T - 1)
        # what every WAVE of the run loads first (a later run's wave 0: the predecessor) rides in the record itself
        waves = ([] if first else [r["pred_unit"]]) + list(ids)
        for w in range(16):
            ui = waves[w] if w < len(waves) else 0
            assert r["wave_unit"][w] == ui and r["wave_coef"][w] == wl["units"][ui]["coef_offset"] and r["wave_meta"][w] == wl["units"][ui]["meta_offset"]
            assert (r["wave_nch"] >> (2 * w)) & 3 == wl["units"][ui]["n_ch"]
    assert (seen == 1).all()
    # consecutive runs of one chain sit a multiple of 8 blocks apart (same XCD under the observed b % 8 dispatch)
    pos = {int(r["unit"][0]): i for i, r in enumerate(runs)}
    assert sum((pos[s * T + 16] - pos[s * T]) % 8 == 0 for s in range(5)) >= 3


def test_planner_rejects_bad_input(emu):
    wl = _workload(n_streams=1, n_frames=3)
    u = wl["units"].copy()
    u["ch"]["window_sequence"][1, 0] = 2                       # short with long grouping
    assert emu.plan(u, 1, 2)[0] == -1
    u = wl["units"].copy()
    u["n_ch"][2] = 1                                           # layout change mid-batch
    assert emu.plan(u, 1, 2)[0] == -6
    u = wl["units"].copy()
    u["stream"][0] = 7
    assert emu.plan(u, 1, 2)[0] == -4
    u = wl["units"].copy()
    u["ch"]["max_sfb"][0, 1] = 50
    assert emu.plan(u, 1, 2)[0] == -1


def test_uncovered_channels_are_zero(emu):
    """decoder.js:229-231: channels no element writes stay zero."""
    wl = _workload(n_streams=1, n_frames=2, layout=("sce",))
    u = wl["units"].copy()
    u["n_out_ch"] = 2
    u["pcm_offset"] = np.arange(2) * 2048
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    pcm = emu.decode(u, wl["q"], wl["meta"], 4096, pool, par).reshape(2, 1024, 2)
    assert (pcm[:, :, 1] == 0).all() and np.abs(pcm[:, :, 0]).max() > 0


def test_ragged_streams_and_eight_channels(emu, oracle):
    """Chains of different lengths in one batch (1, 3, 17, 33 frames: up to three runs) and an 8-channel layout."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    counts = [3, 17, 1, 33]
    parts = [aacgpu_workload.make_batch(n_streams=1, n_frames=T, mix=True, intensity=True, seed=50 + s, stream_base=s)
             for s, T in enumerate(counts)]
    units = np.concatenate([p["units"] for p in parts])
    q = np.concatenate([p["q"] for p in parts])
    meta = np.concatenate([p["meta"] for p in parts])
    off = np.concatenate([np.full(c, b) for c, b in zip(counts, np.cumsum([0] + counts[:-1]))])
    units["pcm_offset"] += (off * 2048).astype(np.uint32)
    units["coef_offset"] += (off * 2).astype(np.uint32)
    units["meta_offset"] += (off * 2).astype(np.uint32)
    n_pcm = sum(counts) * 2048
    ov = np.zeros((4, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, q, meta, n_pcm, ov)
    pool = np.zeros((4, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(8, np.uint8)
    pcm = emu.decode(units, q, meta, n_pcm, pool, par)
    assert rms(pcm, ref) < RMS_TOL
    assert np.abs(emu_lib.pool_current(pool, par) - ov).max() < 1e-5 * max(1.0, np.abs(ov).max())

    wl = aacgpu_workload.make_batch(n_streams=1, n_frames=3, layout=("cpe", "sce", "cpe", "cpe", "sce"), mix=True, seed=77)
    ov = np.zeros((1, 8, 1024), np.float32)
    ref = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
    pool = np.zeros((1, 8, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(8, np.uint8)
    pcm = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    assert rms(pcm, ref) < RMS_TOL


@pytest.mark.parametrize("layout,T", [(("cpe", "cpe", "cpe", "sce"), 40), (("sce", "cpe", "cpe", "sce"), 21), (("sce", "cpe", "cpe", "cpe", "sce"), 7),
                                      (("sce",) * 8, 5)])
def test_multichannel_long_chains(emu, oracle, layout, T):
    """Multichannel layouts (7 channels, 5.1, 7.1, eight mono elements) with chains longer than a run: their PCM is
    overlap-added in place in the previous wave's slot and stored with consecutive samples in consecutive lanes; later runs
    redo the frame before them; two consecutive batches chain through the overlap state."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    S = 2
    C = sum(2 if e == "cpe" else 1 for e in layout)
    ov = np.zeros((S, C, 1024), np.float32)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    for batch in range(2):
        wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=900 + batch, frame_base=batch * T)
        ref = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
        pcm = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
        assert rms(pcm, ref) < RMS_TOL
        assert np.abs(emu_lib.pool_current(pool, par) - ov).max() < 1e-5 * max(1.0, np.abs(ov).max())


def pcm16(ref):
    """What AACG_OUTPUT_I16 means: round-to-nearest-even of x * 32768, saturated."""
    return np.clip(np.rint(ref.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("layout,T", [(("cpe",), 20), (("cpe", "cpe", "cpe", "sce"), 5), (("sce",), 18)])
def test_int16_output(emu, oracle, layout, T):
    """AACG_OUTPUT_I16: the same samples as int16.  Against the oracle's float PCM rounded the same way: the float results
    differ by ~1e-6 of their magnitude (a few hundredths of a step near full scale), so a sample next to a rounding boundary may
    land one step off — never more, and for under 1 % of the samples."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    S = 2
    C = sum(2 if e == "cpe" else 1 for e in layout)
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=31)
    ov = np.zeros((S, C, 1024), np.float32)
    want = pcm16(oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov))
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    got = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, np.zeros(S * C, np.uint8), int16_out=True)
    d = got.astype(np.int32) - want
    assert np.abs(d).max() <= 1 and np.count_nonzero(d) <= 1e-2 * d.size, (np.abs(d).max(), np.count_nonzero(d))
    assert np.abs(want).max() > 1000


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_vs_oracle(emu, oracle, seed):
    """Random layouts, sequences, shapes (also previous shapes), groupings, band types, masks: emulated kernels vs oracle."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    wl = aacgpu_workload.random_batch(seed, n_streams=2, max_frames=6)
    S, C = wl["n_streams"], wl["max_channels"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref, spec_ref = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    pcm = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    rms(pcm, ref)
    assert np.abs(emu_lib.pool_current(pool, par) - ov).max() < 1e-5 * max(1.0, np.abs(ov).max())
    spec = emu.spectral(wl["units"], wl["q"], wl["meta"])
    assert np.array_equal(spec.view(np.uint32), spec_ref.view(np.uint32))


@pytest.mark.parametrize("T,want", [(31, [16, 15]), (32, [16, 16]), (33, [16, 15, 2]), (48, [16, 16, 16])])
def test_long_chains_double_duty(emu, oracle, T, want):
    """Chains longer than a run: a later run recomputes its predecessor's tail — in a wave of its own when it
    holds up to 15 frames, by its first wave doing double duty when a full 16 keeps the number of runs minimal."""
    wl = _workload(n_streams=1, n_frames=T, mix=True, intensity=True, seed=T)
    n, runs, info = emu.plan(wl["units"], 1, 2)
    assert sorted((int(r["n_units"]) for r in runs), reverse=True) == want
    assert sum(int(r["pred_unit"]) >= 0 for r in runs) == len(want) - 1
    ov = np.zeros((1, 2, 1024), np.float32)
    ref = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    pcm = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    assert rms(pcm, ref) < RMS_TOL
    assert np.abs(emu_lib.pool_current(pool, par) - ov).max() < 1e-5 * max(1.0, np.abs(ov).max())
    # f32 seam, single channel element as well
    wl = _workload(n_streams=1, n_frames=T, layout=("sce",), mix=True, seed=T + 1)
    ov = np.zeros((1, 1, 1024), np.float32)
    ref, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    pool = np.zeros((1, 1, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(1, np.uint8)
    pcm = emu.decode(wl["units"], spec, None, wl["n_pcm"], pool, par)
    assert rms(pcm, ref) < RMS_TOL


@pytest.mark.parametrize("sample_index,max_long", [(5, 49), (6, 47), (8, 43), (0, 41)])
def test_other_sample_rates_vs_oracle(emu, oracle, sample_index, max_long):
    """The band tables depend on the sampling rate (tables.js:34-155): 32, 24, 16 and 96 kHz layouts against the
    oracle (the golden vectors pin 48 kHz).  The generator's long windows are clamped to the rate's band count."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aac.js_amd", "python"))
    import aacgpu_workload
    wl = aacgpu_workload.random_batch(500 + sample_index, n_streams=2, max_frames=5)
    units = wl["units"].copy()
    for i in range(len(units)):
        for c in range(2):
            short = int(units[i]["ch"][c]["window_sequence"]) == 2
            units[i]["ch"][c]["max_sfb"] = min(int(units[i]["ch"][c]["max_sfb"]), 12 if short else max_long)
    # grouped shorts index their bands by g * maxSFB + sfb: rebuild nothing, the oracle and the kernels read the
    # same words through the same formula
    S, C = wl["n_streams"], wl["max_channels"]
    ov = np.zeros((S, C, 1024), np.float32)
    ref, spec_ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, sample_index=sample_index, want_spec=True)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    pcm = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, sample_index=sample_index)
    rms(pcm, ref)
    spec = emu.spectral(units, wl["q"], wl["meta"], sample_index=sample_index)
    assert np.array_equal(spec.view(np.uint32), spec_ref.view(np.uint32))
