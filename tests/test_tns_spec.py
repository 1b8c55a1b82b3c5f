"""AACG_TNS_SPEC: the all-pole filter tns.js:105-177 was meant to run.

aac.js itself never executes it (NaN loop bounds, tns.js:106,122), so there is no reference output to
pin to: PARITY UNPINNED BY THE REFERENCE for this mode.  The oracle's restatement (orc_tns_spec) is
instead checked against an independent double-precision direct form written from ISO/IEC 14496-3
4.6.9.3 below, and the kernels (block scan, different summation order) against the oracle within a
stated tolerance.  The default mode (AACG_TNS_REFERENCE) must ignore TNS side info bit for bit.
"""
import numpy as np
import pytest

import aacgpu_workload as W
import emu_lib
import orc

# 48 kHz tables (ISO/IEC 14496-3 Tables 4.129/4.130, 4.138)
TNS_MAX_BANDS = {False: 40, True: 14}

# Filtered spectra vs the float64 direct form.  The recursion amplifies the float32 store rounding (and the
# float32 rounding of the LPC coefficients, tns.js:39) by up to the square of the filter's gain, and random
# reflection coefficients of order 12 reach gains of 30+: the forward bound scales accordingly, and the
# well-conditioned statement (the output satisfies the recurrence to float32 rounding) is checked separately.
ORACLE_REL_TOL = 3e-6
# Kernel (block scan) vs oracle (sample-serial), relative to the PCM signal RMS.  Inside a 16-sample block the
# kernel uses the oracle's arithmetic; the state entering a block comes from the scan (block transitions in
# double precision), so it differs from the oracle's float32-rounded history by that rounding times the filter's
# gain.  Measured: ~1.5e-6 (the IMDCT's own level) with encoder-like filters; with near-unstable random filters
# <= 3e-5, where the oracle itself is up to 7e-4 from the float64 evaluation.
KERNEL_REL_TOL = 5e-6
KERNEL_REL_TOL_WILD = 1e-3     # filters no encoder emits (PCM peaks of 1e4 on the [-1,1) scale): add_tns(wild=True)


def direct_form(info, rec, data, want_filters=False):
    """ISO/IEC 14496-3 4.6.9.3 tns_decode_frame in float64, one channel, spectrum in ICStream.data order."""
    short = int(info["window_sequence"]) == 2
    swb = W.SWB_SHORT_48 if short else W.SWB_LONG_48
    n_swb = len(swb) - 1
    mmm = min(TNS_MAX_BANDS[short], int(info["max_sfb"]))
    out = np.array(data, np.float64)
    filters = []
    for w in range(8 if short else 1):
        bottom = n_swb
        for f in range(int(rec["n_filt"][w])):
            flt = rec["filt"][w if short else f]
            top = bottom
            bottom = max(top - int(flt["length"]), 0)
            order = int(flt["order"])
            if order == 0:
                continue
            # tns_decode_coef: reflection -> direct form (the tables of tns.js:49-63 hold -sin(), hence the sign)
            k = -flt["coef"][:order].astype(np.float64)
            a = np.zeros(order + 1)
            a[0] = 1.0
            for m in range(1, order + 1):
                b = a.copy()
                for i in range(1, m):
                    b[i] = a[i] + k[m - 1] * a[m - i]
                b[m] = k[m - 1]
                a = b
            start, end = int(swb[min(bottom, mmm)]), int(swb[min(top, mmm)])
            size = end - start
            if size <= 0:
                continue
            inc = 1
            if int(flt["direction"]):
                inc, start = -1, end - 1
            pos = w * 128 + start
            filters.append((pos, size, inc, a))
            hist = []
            for m in range(size):
                y = out[pos]
                for j in range(1, min(m, order) + 1):
                    y -= a[j] * hist[-j]
                out[pos] = y
                hist.append(y)
                pos += inc
    return (out, filters) if want_filters else out


def _random_channel(rng, short):
    info = np.zeros((), orc.CHAN_INFO_DTYPE)
    info["window_sequence"] = 2 if short else int(rng.choice([0, 1, 3]))
    info["max_sfb"] = int(rng.integers(1, 15)) if short else int(rng.integers(1, 50))
    info["group_count"] = 1
    info["group_len"][0] = 8 if short else 1
    unit = np.zeros(1, orc.UNIT_DTYPE)
    unit["n_ch"] = 1
    unit["ch"][0][0] = info
    return info, unit


@pytest.mark.parametrize("short", [False, True])
def test_oracle_tns_matches_iso_direct_form(oracle, short):
    rng = np.random.default_rng(77 + short)
    worst = 0.0
    for trial in range(40):
        info, unit = _random_channel(rng, short)
        units, tns = W.add_tns(dict(units=unit), seed=1000 * short + trial, p_channel=1.0, wild=trial % 2 == 1)
        data = (rng.standard_normal(1024) * 1000).astype(np.float32)
        got = oracle.tns_spec(info, tns[0], data)
        want, filters = direct_form(info, tns[0], data, want_filters=True)
        err = np.sqrt(np.mean((got - want) ** 2))
        sig = np.sqrt(np.mean(want ** 2))
        gain = max(1.0, sig / np.sqrt(np.mean(data.astype(np.float64) ** 2)))
        worst = max(worst, err / sig / gain ** 2)
        # samples outside every filter's range are untouched, bit for bit
        same = want == data.astype(np.float64)
        assert np.array_equal(got[same].view(np.uint32), data[same].view(np.uint32))
        # backward check: x[m] = y[m] + sum a[k] y[m-k] holds to a few float32 roundings of the terms
        y = got.astype(np.float64)
        for pos, size, inc, a in filters:
            idx = pos + inc * np.arange(size)
            for m in range(size):
                k = np.arange(1, min(m, len(a) - 1) + 1)
                terms = a[k] * y[idx[m - k]]
                resid = y[idx[m]] + terms.sum() - np.float64(data[idx[m]])
                bound = 2.0 ** -23 * (abs(y[idx[m]]) + np.abs(terms).sum()) * 2 + 1e-30
                assert abs(resid) <= bound, (trial, m, resid, bound)
    assert worst < ORACLE_REL_TOL, worst


def test_oracle_tns_edge_cases(oracle):
    info, unit = _random_channel(np.random.default_rng(1), False)
    info["max_sfb"] = 49
    data = np.arange(1024, dtype=np.float32) - 300
    rec = np.zeros((), orc.TNS_DTYPE)
    # no filters / order 0 / zero length: identity
    assert np.array_equal(oracle.tns_spec(info, rec, data), data)
    rec["n_filt"][0] = 2
    rec["filt"][0]["length"] = 20
    rec["filt"][1]["length"] = 0
    rec["filt"][1]["order"] = 5
    rec["filt"][1]["coef"][:5] = 0.5
    assert np.array_equal(oracle.tns_spec(info, rec, data), data)
    # order-1 filter, upwards over the top 9 bands (40 = TNS_MAX_BANDS caps `top`): y[m] = x[m] + c y[m-1]
    rec = np.zeros((), orc.TNS_DTYPE)
    rec["n_filt"][0] = 1
    rec["filt"][0]["length"] = 18           # bands 31..48 -> clipped to 31..39
    rec["filt"][0]["order"] = 1
    rec["filt"][0]["coef"][0] = 0.5         # lpc[0] = -0.5
    got = oracle.tns_spec(info, rec, data)
    lo, hi = int(W.SWB_LONG_48[31]), int(W.SWB_LONG_48[40])
    want = data.copy()
    for n in range(lo + 1, hi):
        want[n] = np.float32(np.float64(data[n]) + 0.5 * np.float64(want[n - 1]))
    assert np.array_equal(got, want)
    # downwards: the mirror image
    rec["filt"][0]["direction"] = 1
    got = oracle.tns_spec(info, rec, data)
    want = data.copy()
    for n in range(hi - 2, lo - 1, -1):
        want[n] = np.float32(np.float64(data[n]) + 0.5 * np.float64(want[n + 1]))
    assert np.array_equal(got, want)


def _decode_pair(oracle, wl, tns, units=None):
    units = wl["units"] if units is None else units
    S = wl.get("n_streams", int(units["stream"].max()) + 1)
    C = wl.get("max_channels", wl.get("C"))
    ov = np.zeros((S, C, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)
    return ref, ov, S, C


def test_reference_mode_ignores_tns_side_info(oracle):
    wl = W.random_batch(5, n_streams=2, max_frames=4)
    units, tns = W.add_tns(wl, seed=5)
    plain, ov0, S, C = _decode_pair(oracle, wl, None)
    flagged, ov1, _, _ = _decode_pair(oracle, wl, None, units=units)     # flags set, REFERENCE mode
    assert np.array_equal(plain.view(np.uint32), flagged.view(np.uint32))
    spec, _, _, _ = _decode_pair(oracle, wl, tns, units=units)
    assert not np.array_equal(plain, spec)                               # SPEC mode does change the output


@pytest.fixture(scope="module")
def emu():
    return emu_lib.Emu()


def _rel(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b * b)), 1e-30))


@pytest.mark.parametrize("seed,wild", [(11, False), (12, False), (13, False), (13, True)])
def test_emulated_kernels_tns_vs_oracle(emu, oracle, seed, wild):
    """Random layouts / sequences with TNS on ~60 % of the channels: block-scan kernels vs the oracle."""
    wl = W.random_batch(seed, n_streams=2, max_frames=5)
    units, tns = W.add_tns(wl, seed=seed, wild=wild)
    ref, ov, S, C = _decode_pair(oracle, wl, tns, units=units)
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    pcm = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns)
    tol = KERNEL_REL_TOL_WILD if wild else KERNEL_REL_TOL
    assert _rel(pcm, ref) < tol
    assert _rel(emu_lib.pool_current(pool, par), ov) < tol
    # without records the flags are inert (REFERENCE behaviour), bit-identical to the unflagged batch
    pool[:] = 0
    par[:] = 0
    a = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    pool[:] = 0
    par[:] = 0
    b = emu.decode(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], pool, par)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_emulated_kernels_tns_f32_seam(emu, oracle):
    """SPEC_F32 input (filterbank seam): TNS runs on the caller's spectrum."""
    wl = W.make_batch(n_streams=2, n_frames=5, mix=True, seed=99)
    units, tns = W.add_tns(wl, seed=3, p_channel=0.8)
    ov = np.zeros((2, 2, 1024), np.float32)
    _, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    ov[:] = 0
    ref = oracle.decode_batch(units, spec, None, wl["n_pcm"], ov, tns=tns)
    ov2 = np.zeros_like(ov)
    ref_q = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov2, tns=tns)
    assert np.array_equal(ref.view(np.uint32), ref_q.view(np.uint32))    # both seams agree in the oracle
    pool = np.zeros((2, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(4, np.uint8)
    pcm = emu.decode(units, spec, None, wl["n_pcm"], pool, par, tns=tns)
    assert _rel(pcm, ref) < KERNEL_REL_TOL


@pytest.mark.parametrize("seed", [11, 14])
def test_emulated_one_launch_and_staged_routes_agree(emu, oracle, seed):
    """The engine runs the filters inside the run kernel (f32 PCM, no coupling elements) or as a launch of their own
    (int16 PCM, coupling): the same arithmetic either way, so the same bits; and the int16 route rounds that PCM."""
    wl = W.random_batch(seed, n_streams=2, max_frames=5)
    units, tns = W.add_tns(wl, seed=seed)
    S, C = int(units["stream"].max()) + 1, 8
    out = []
    for staged in (False, True):
        pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
        par = np.zeros(S * C, np.uint8)
        out.append((emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns, staged=staged), pool.copy()))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    pool = np.zeros((S, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(S * C, np.uint8)
    p16 = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns, int16_out=True)
    want = np.clip(np.rint(out[0][0].astype(np.float64) * 32768.0), -32768, 32767).astype(np.int32)
    ok = ~np.isnan(out[0][0])
    assert np.abs(p16.astype(np.int32)[ok] - want[ok]).max() <= 1


def test_planner_rejects_bad_tns(emu):
    wl = W.make_batch(n_streams=1, n_frames=2, seed=1)
    units, tns = W.add_tns(wl, seed=1, p_channel=1.0)
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    bad = units.copy()
    bad["tns_offset"][1] = len(tns)                   # points outside the array
    with pytest.raises(RuntimeError, match="tns_offset"):
        emu.decode(bad, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns)
    t2 = tns.copy()
    t2[0]["n_filt"][0] = 1
    t2[0]["filt"][0]["order"] = 13                     # above the AAC-LC limit
    with pytest.raises(RuntimeError, match="rc=-5"):
        emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=t2)


def test_refresh_of_a_kept_plan_refuses_tns_frames_on_a_spec_engine(emu):
    """a plan made from a batch without TNS must not take a later batch WITH TNS side info on an AACG_TNS_SPEC
    engine (the refresh would clear the flags and decode without the filters); a REFERENCE engine ignores the side info."""
    wl = W.make_batch(n_streams=2, n_frames=4, seed=5)
    with_tns, _ = W.add_tns(wl, seed=5, p_channel=1.0)
    assert emu.plan_refresh(wl["units"], wl["units"], 2, 2, tns_spec=True) == 0
    assert emu.plan_refresh(wl["units"], with_tns, 2, 2, tns_spec=False) == 0
    assert emu.plan_refresh(wl["units"], with_tns, 2, 2, tns_spec=True) == -6          # AACG_ERR_LAYOUT_CHANGE
    assert "TNS" in emu.error()


# ---- the HIP path (needs a real MI355X) -----------------------------------------------------------------
def _gpu_overlaps(eng, S, C):
    return np.stack([[eng.get_overlap(s, c) for c in range(C)] for s in range(S)])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,wild", [(201, False), (202, False), (203, False), (204, False), (205, True), (206, True)])
def test_gpu_tns_fuzz_vs_oracle(oracle, seed, wild):
    import aacgpu
    wl = W.random_batch(seed, n_streams=4, max_frames=20)
    units, tns = W.add_tns(wl, seed=seed, wild=wild)
    ref, ov, S, C = _decode_pair(oracle, wl, tns, units=units)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C, tns_mode=aacgpu.TNS_SPEC)
    pcm = eng.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], tns=tns)
    tol = KERNEL_REL_TOL_WILD if wild else KERNEL_REL_TOL
    assert _rel(pcm, ref) < tol
    assert _rel(_gpu_overlaps(eng, S, C), ov) < tol
    eng.close()
    # REFERENCE-mode engine: side info and flags are ignored, bit-identical to the plain batch
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C)
    a = eng.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], tns=tns)
    eng.close()
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C)
    b = eng.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"])
    eng.close()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("inp", ["q", "f"])
def test_gpu_tns_int16_engine_takes_the_staged_route(oracle, inp):
    """AACG_OUTPUT_I16 + AACG_TNS_SPEC: the filters as a launch of their own in front of the int16 run kernel; the samples are
    the one-launch engine's f32 PCM, rounded (at most one step apart at a rounding boundary)."""
    import aacgpu
    wl = W.random_batch(207, n_streams=4, max_frames=20)
    units, tns = W.add_tns(wl, seed=207)
    ref, ov, S, C = _decode_pair(oracle, wl, tns, units=units)
    kind = aacgpu.INPUT_QUANT_I16 if inp == "q" else aacgpu.INPUT_SPEC_F32
    coeffs, meta = wl["q"], wl["meta"]
    if inp == "f":
        ov0 = np.zeros((S, C, 1024), np.float32)
        _, coeffs = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov0, want_spec=True)   # spectra before TNS
        meta = None
    eng = aacgpu.Engine(kind, S, C, tns_mode=aacgpu.TNS_SPEC)
    f32 = eng.decode_batch(units, coeffs, meta, wl["n_pcm"], tns=tns)
    eng.close()
    assert _rel(f32, ref) < KERNEL_REL_TOL
    eng = aacgpu.Engine(kind, S, C, tns_mode=aacgpu.TNS_SPEC, output_kind=aacgpu.OUTPUT_I16)
    i16 = eng.decode_batch(units, coeffs, meta, wl["n_pcm"], tns=tns)
    eng.close()
    assert i16.dtype == np.int16
    want = np.clip(np.rint(f32.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int32)
    d = np.abs(i16.astype(np.int32) - want)
    assert d.max() <= 1 and np.count_nonzero(d) <= 1e-2 * d.size


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [("cpe",), ("cpe", "cpe", "cpe", "sce")])
def test_gpu_tns_cfg3_both_seams(oracle, layout):
    """BASELINE config 3 ("mixed window sequences, TNS on") at reduced size, through both input seams and
    through the device-resident plan path."""
    import aacgpu
    import torch
    S, T = 6, 19
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=77)
    C = wl["C"]
    units, tns = W.add_tns(wl, seed=7, p_channel=0.7)
    ov = np.zeros((S, C, 1024), np.float32)
    _, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    ov[:] = 0
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)

    eng = aacgpu.Engine(aacgpu.INPUT_SPEC_F32, S, C, tns_mode=aacgpu.TNS_SPEC)
    pcm = eng.decode_batch(units, spec, None, wl["n_pcm"], tns=tns)
    assert _rel(pcm, ref) < KERNEL_REL_TOL
    assert _rel(_gpu_overlaps(eng, S, C), ov) < KERNEL_REL_TOL
    eng.close()

    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, C, tns_mode=aacgpu.TNS_SPEC)
    plan = eng.plan(units, tns=tns)
    dq = torch.from_numpy(wl["q"]).cuda()
    dm = torch.from_numpy(wl["meta"].view(np.int16)).cuda()
    dp = torch.zeros(wl["n_pcm"], dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.decode_device(plan, dq.data_ptr(), dm.data_ptr(), dp.data_ptr(), 0)
    eng.synchronize()
    assert _rel(dp.cpu().numpy(), ref) < KERNEL_REL_TOL
    assert _rel(_gpu_overlaps(eng, S, C), ov) < KERNEL_REL_TOL
    plan.destroy()
    eng.close()


@pytest.mark.gpu
def test_gpu_tns_errors():
    import aacgpu
    wl = W.make_batch(n_streams=1, n_frames=2, seed=1)
    units, tns = W.add_tns(wl, seed=1, p_channel=1.0)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, 1, 2, tns_mode=aacgpu.TNS_SPEC)
    t2 = tns.copy()
    t2[0]["n_filt"][0] = 1
    t2[0]["filt"][0]["order"] = 13
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], tns=t2)
    assert ei.value.code == -5
    bad = units.copy()
    bad["tns_offset"][1] = len(tns)
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.decode_batch(bad, wl["q"], wl["meta"], wl["n_pcm"], tns=tns)
    assert ei.value.code == -1
    # flags without records: nothing to apply, decodes like a plain batch
    a = eng.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"])
    eng.reset_stream(0)
    b = eng.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    eng.close()
    with pytest.raises(aacgpu.AacgError):
        aacgpu.Engine(aacgpu.INPUT_QUANT_I16, 1, 2, tns_mode=2)
    # a kept plan without TNS records refuses a refresh with TNS frames on this engine
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, 1, 2, tns_mode=aacgpu.TNS_SPEC)
    plan = eng.plan(wl["units"])
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.plan_refresh_units(plan, units)
    assert ei.value.code == -6
    eng.plan_refresh_units(plan, wl["units"])
    plan.destroy()
    eng.close()


def test_emulated_tns_long_chain(emu, oracle):
    """A chain longer than a run with TNS side info: the TNS stage is a kernel of its own, so the run kernel's
    double-duty variant (16 + 16 frames) applies as for any other batch."""
    wl = W.make_batch(n_streams=1, n_frames=32, mix=True, seed=5)
    units, tns = W.add_tns(wl, seed=9)
    ov = np.zeros((1, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)
    pool = np.zeros((1, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(2, np.uint8)
    pcm = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns)
    assert _rel(pcm, ref) < KERNEL_REL_TOL
    assert _rel(emu_lib.pool_current(pool, par), ov) < KERNEL_REL_TOL


@pytest.mark.gpu
def test_gpu_tns_long_chain(oracle):
    import aacgpu
    S, T = 3, 40
    wl = W.make_batch(n_streams=S, n_frames=T, mix=True, intensity=True, seed=6)
    units, tns = W.add_tns(wl, seed=10)
    ov = np.zeros((S, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, S, 2, tns_mode=aacgpu.TNS_SPEC)
    pcm = eng.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], tns=tns)
    assert _rel(pcm, ref) < KERNEL_REL_TOL
    assert _rel(_gpu_overlaps(eng, S, 2), ov) < KERNEL_REL_TOL
    eng.close()


def test_config3_tns_workload(emu, oracle):
    """a filter on every channel-frame (long: order 12 over 20 bands; short: order 7 per window)."""
    wl = W.make_batch(n_streams=2, n_frames=9, mix=True, seed=3)
    units, tns = W.add_tns_config3(wl)
    assert len(tns) == 2 * len(units) and all(int(u["ch"][c]["flags"]) & 1 for u in units for c in range(2))
    ov = np.zeros((2, 2, 1024), np.float32)
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)
    pool = np.zeros((2, 2, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(4, np.uint8)
    pcm = emu.decode(units, wl["q"], wl["meta"], wl["n_pcm"], pool, par, tns=tns)
    assert _rel(pcm, ref) < KERNEL_REL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("seam,S,T", [("q", 128, 16), ("f", 48, 40)])
def test_gpu_tns_launches_overlapped_equal_launch_behind_launch(oracle, seam, S, T):
    """AACG_TNS_SPEC batches through aacg_decode_pipelined (aacg_imdct_run_*_ex_rv: the filters inside the run kernel, chains that
    meet in rendezvous cells — between the runs of a chain longer than 16 frames and between consecutive launches): 40 overlapped
    launches of one plan against the same launches one behind the other, np.array_equal on the bits, and the overlap state; the
    first launch against the oracle."""
    import aacgpu
    import torch
    n = 40
    wl = W.make_batch(n_streams=S, n_frames=T, layout=("cpe",), mix=True, intensity=True, seed=88)
    C = wl["C"]
    units, tns = W.add_tns(wl, seed=8, p_channel=0.7)
    ov = np.zeros((S, C, 1024), np.float32)
    _, spec = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)      # spectra before TNS
    ov[:] = 0
    ref = oracle.decode_batch(units, wl["q"], wl["meta"], wl["n_pcm"], ov, tns=tns)
    kind = aacgpu.INPUT_QUANT_I16 if seam == "q" else aacgpu.INPUT_SPEC_F32
    host = wl["q"] if seam == "q" else spec.astype(np.float32)
    ins = [torch.from_numpy(np.ascontiguousarray(np.roll(host, 97 * j, axis=0))).cuda() for j in range(n)]
    dm = torch.from_numpy(wl["meta"].view(np.int16)).cuda() if seam == "q" else None
    mp = dm.data_ptr() if dm is not None else None
    results = []
    for pipelined in (False, True):
        eng = aacgpu.Engine(kind, S, C, tns_mode=aacgpu.TNS_SPEC)
        plan = eng.plan(units, tns=tns)
        name = eng.plan_kernels(plan, pipelined=pipelined)
        which = "quant" if seam == "q" else "f32"
        assert name == ("aacg_imdct_run_%s_ex_rv" % which if pipelined or T > 16 else "aacg_imdct_run_%s_ex" % which)
        outs = [torch.full((wl["n_pcm"],), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        for j in range(n):
            if pipelined:
                eng.decode_pipelined(plan, ins[j].data_ptr(), mp, outs[j].data_ptr())
            else:
                eng.decode_device(plan, ins[j].data_ptr(), mp, outs[j].data_ptr(), 0)
        eng.synchronize()
        torch.cuda.synchronize()
        if pipelined:
            assert eng.pipeline_chained() == n - 1
        results.append(([o.cpu().numpy() for o in outs], _gpu_overlaps(eng, S, C)))
        plan.destroy()
        eng.close()
    (serial, s_state), (piped, p_state) = results
    assert _rel(serial[0], ref) < KERNEL_REL_TOL
    for j in range(n):
        assert not np.isnan(piped[j]).any(), j
        assert np.array_equal(piped[j].view(np.uint32), serial[j].view(np.uint32)), "launch %d differs from the serialised route" % j
    assert np.array_equal(p_state.view(np.uint32), s_state.view(np.uint32))
