"""The oracle (oracle/aac_oracle.c) must reproduce the REFERENCE's own outputs bit-for-bit.

Golden vectors were produced by running audiocogs/aac.js under Node (tests/golden/gen/gen_golden.js);
the reference ships no tests of its own, so these are the pins.  Bit-exact: compared
as uint32 views, so -0.0 vs +0.0 and NaN payloads count too.
"""
import numpy as np
import pytest

import orc


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what):
    a = np.ascontiguousarray(a, np.float32).ravel()
    b = np.ascontiguousarray(b, np.float32).ravel()
    assert a.shape == b.shape, what
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, "%s: %d/%d floats differ, first at %d: %r vs %r" % (
        what, bad.size, a.size, bad[0], a[bad[0]], b[bad[0]])


@pytest.mark.parametrize("which,name", [(0, "tables.iq"), (1, "tables.sf"), (2, "tables.sine_long"),
                                        (3, "tables.kbd_long"), (4, "tables.sine_short"), (5, "tables.kbd_short"),
                                        (6, "tables.fft_roots_512"), (7, "tables.fft_roots_64")])
def test_tables_f32(oracle, golden, which, name):
    assert_bit_equal(oracle.table_f32(which), golden[name], name)


@pytest.mark.parametrize("which,name", [(0, "tables.mdct_2048"), (1, "tables.mdct_256")])
def test_tables_mdct(oracle, golden, which, name):
    got = oracle.table_f64(which)
    ref = golden[name].ravel()
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), name


def test_swb_offsets(oracle, golden):
    counts = golden["tables.swb_counts"]
    for s in range(12):
        for is_long, tab, row in ((1, "tables.swb_long", 0), (0, "tables.swb_short", 1)):
            n = int(counts[row, s])
            got = oracle.swb_offsets(s, is_long)
            assert len(got) == n + 1
            assert np.array_equal(got, golden[tab][s, :n + 1]), (s, is_long)


@pytest.mark.parametrize("n", [512, 64])
def test_fft(oracle, golden, n):
    xin, xout = golden["fft%d.in" % n], golden["fft%d.out" % n]
    for v in range(xin.shape[0]):
        assert_bit_equal(oracle.fft_inverse(xin[v]), xout[v], "fft%d[%d]" % (n, v))


@pytest.mark.parametrize("n", [2048, 256])
def test_imdct(oracle, golden, n):
    xin, xout = golden["imdct%d.in" % n], golden["imdct%d.out" % n]
    for v in range(xin.shape[0]):
        assert_bit_equal(oracle.imdct(xin[v]), xout[v], "imdct%d[%d]" % (n, v))


def test_imdct_matches_definition(oracle, golden):
    """y[n] = (2/N) sum X[k] cos(2pi/N (n + n0)(k + 1/2)), n0 = (N/2+1)/2."""
    for N in (2048, 256):
        x = golden["imdct%d.in" % N][0].astype(np.float64)
        n = np.arange(N)[:, None]
        k = np.arange(N // 2)[None, :]
        y = (2.0 / N) * (np.cos(2 * np.pi / N * (n + (N / 2 + 1) / 2) * (k + 0.5)) @ x)
        got = oracle.imdct(x.astype(np.float32)).astype(np.float64)
        rel = np.sqrt(np.mean((got - y) ** 2) / np.mean(y ** 2))
        assert rel < 5e-6, (N, rel)


def test_pns_sequence_known_bad(oracle, golden):
    """ics.js:234 as written: 11 draws, then zeros forever."""
    seq = oracle.pns_sequence(16)
    assert np.array_equal(seq, golden["pns.sequence"])
    assert seq[10] == -2147483648 and not seq[11:].any()


def test_cfg1_mono_long(oracle, golden):
    """BASELINE config 1: one mono ONLY_LONG sine frame through readChunk."""
    units = np.zeros(1, orc.UNIT_DTYPE)
    units["n_out_ch"] = 1
    units["n_ch"] = 1
    units["ch"]["max_sfb"][0, 0] = 49
    units["ch"]["group_count"][0, 0] = 1
    units["ch"]["group_len"][0, 0, 0] = 1
    ov = np.zeros((1, 1, 1024), np.float32)
    pcm = oracle.decode_batch(units, golden["cfg1.spec"], None, 1024, ov)
    assert_bit_equal(pcm, golden["cfg1.pcm"], "cfg1 pcm")
    assert_bit_equal(ov, golden["cfg1.overlap"], "cfg1 overlap")


SCENARIOS = ["scn_stereo", "scn_split", "scn_7ch", "scn_mono"]


@pytest.mark.parametrize("name", SCENARIOS)
def test_scenario_quant(oracle, golden, name):
    """process(elements) seam: dequant + MS + IS + filterbank + interleave, all frames chained."""
    units = golden[name + ".units"].view(orc.UNIT_DTYPE).ravel()
    pcm_ref = golden[name + ".pcm"]
    C = pcm_ref.shape[2]
    ov = np.zeros((1, C, 1024), np.float32)
    pcm, spec = oracle.decode_batch(units, golden[name + ".q"], golden[name + ".meta"], pcm_ref.size, ov,
                                    want_spec=True)
    assert_bit_equal(spec, golden[name + ".spec"], name + " spectrum after MS/IS")
    assert_bit_equal(pcm, pcm_ref, name + " pcm")
    assert_bit_equal(ov, golden[name + ".overlap"], name + " final overlap")


@pytest.mark.parametrize("name", SCENARIOS)
def test_scenario_spec(oracle, golden, name):
    """filterbank seam: the spectra the reference fed to FilterBank.process, as f32 input."""
    units = golden[name + ".units"].view(orc.UNIT_DTYPE).ravel()
    pcm_ref = golden[name + ".pcm"]
    C = pcm_ref.shape[2]
    ov = np.zeros((1, C, 1024), np.float32)
    pcm = oracle.decode_batch(units, golden[name + ".spec"], None, pcm_ref.size, ov)
    assert_bit_equal(pcm, pcm_ref, name + " pcm")
    assert_bit_equal(ov, golden[name + ".overlap"], name + " final overlap")


def test_scenarios_cover_the_cases(golden):
    """The fixtures must actually exercise what they claim: all four sequences, both shapes, MS, IS,
    zero bands, grouped shorts, -0.0 from q == 0, escape-range values."""
    u = golden["scn_stereo.units"].view(orc.UNIT_DTYPE).ravel()
    seqs = set(u["ch"]["window_sequence"][:, 0].tolist())
    assert seqs == {0, 1, 2, 3}
    assert set(u["ch"]["window_shape"][:, 0].tolist()) == {0, 1}
    meta = golden["scn_stereo.meta"]
    bt = meta >> 12
    assert (bt == 15).any() and (bt == 14).any() and (bt == 0).any() and (bt == 11).any()
    assert (meta & 0x400).any()
    assert (u["flags"] & 2).any() and not (u["flags"] & 2).all()
    assert (u["ch"]["group_count"][:, 0] > 1).any()
    spec = golden["scn_stereo.spec"]
    assert (np.signbit(spec) & (spec == 0)).any(), "no -0.0 in fixtures"
    assert np.abs(golden["scn_stereo.q"]).max() > 4000
    u2 = golden["scn_split.units"].view(orc.UNIT_DTYPE).ravel()
    assert (u2["ch"]["window_sequence"][:, 0] != u2["ch"]["window_sequence"][:, 1]).any()


def test_oracle_sanitized(golden):
    """Same scenario under -fsanitize=address,undefined (CPU build only; GPU ASan is unavailable)."""
    import os, subprocess, sys, textwrap
    path = orc.build("liboracle_asan.so")
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import orc, golden_io
        g = golden_io.load_golden()
        o = orc.Oracle(%r)
        for name in ("scn_stereo", "scn_7ch"):
            units = g[name + ".units"].view(orc.UNIT_DTYPE).ravel()
            C = g[name + ".pcm"].shape[2]
            ov = np.zeros((1, C, 1024), np.float32)
            pcm = o.decode_batch(units, g[name + ".q"], g[name + ".meta"], g[name + ".pcm"].size, ov)
            assert np.array_equal(pcm.view(np.uint32), g[name + ".pcm"].ravel().view(np.uint32))
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), path)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_exact_roots_variant_measures_the_recurrence_drift(oracle, golden):
    """the reference's FFT roots come from a float32 recurrence (fft.js:59-103).  With correctly rounded roots
    instead (orc_set_fft_roots) the IMDCT moves by about 1e-6 of its output — the figure BASELINE.md quotes for the
    reference's own distance from an exact transform — and back to bit-exact when the recurrence is restored."""
    x = golden["imdct2048.in"][0]
    want = golden["imdct2048.out"][0]
    y_rec = oracle.imdct(x)
    assert np.array_equal(y_rec.view(np.uint32), want.view(np.uint32))
    with oracle.exact_fft_roots():
        y_exact = oracle.imdct(x)
        k = np.arange(512)
        roots = oracle.table_f32(6).reshape(512, 3)
        assert np.array_equal(roots[:, 0], np.cos(2 * np.pi * k / 512).astype(np.float32))
    rel = float(np.sqrt(np.mean((y_exact.astype(np.float64) - y_rec) ** 2)) / np.sqrt(np.mean(y_rec.astype(np.float64) ** 2)))
    assert 2e-7 < rel < 3e-6, rel
    assert np.array_equal(oracle.imdct(x).view(np.uint32), want.view(np.uint32))          # restored
