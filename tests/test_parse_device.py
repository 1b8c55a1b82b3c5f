"""The device front end (aacg_parser_*, aac.js_amd/csrc/aacg_parse.h): one GPU lane parses one frame.

Expected outputs come from the JavaScript front end (aac.js_amd/js/frontend.js — itself compared with the
reference's parser field by field in tests/js/test_frontend.js) run under Node on streams the synthetic writer
produced (tests/js/parse_cases.js).  Integer / byte work: every comparison is bit-exact.

The cases are written with the standard's codebooks (aac.js_amd/data/aac_codebooks.json; the library's copy,
aacg_standard_codebooks(), must be the same records) and once more with stand-in codebooks
(tests/js/synth_codebooks.js: same alphabets, other prefix codes) — a parser must not care which complete prefix
code it is given.

  not gpu : the kernel source executed lane by lane on the CPU (tests/emu)
  gpu     : aacg_parse_batch on the device; then parse -> aacg_decode_batch against the oracle
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import aacgpu
import emu_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not present")


def make_cases(tmp, mode):
    out = os.path.join(str(tmp), mode)
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "parse_cases.js"), out, mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return out


@pytest.fixture(scope="module")
def standard(tmp_path_factory):
    """Cases written and parsed (JavaScript) with the standard's codebooks."""
    return make_cases(tmp_path_factory.mktemp("parse"), "standard")


@pytest.fixture(scope="module")
def synthetic(tmp_path_factory):
    return make_cases(tmp_path_factory.mktemp("parse_standin"), "synthetic")


def codebooks(d):
    return (np.fromfile(os.path.join(d, "codebooks.entries"), aacgpu.CODE_ENTRY_DTYPE), np.fromfile(os.path.join(d, "codebooks.counts"), np.uint32))


def load_case(d, case):
    f = lambda ext, dt: np.fromfile(os.path.join(d, case["name"] + ext), dt)
    exp = {"units": f(".units", aacgpu.UNIT_DTYPE), "q": f(".q", np.int16).reshape(-1, 1024), "meta": f(".meta", np.uint16).reshape(-1, 120),
           "results": f(".results", aacgpu.PARSE_RESULT_DTYPE), "tns": f(".tns", aacgpu.TNS_DTYPE) if case["wantTns"] else None}
    return f(".bytes", np.uint8), f(".frames", aacgpu.PARSE_FRAME_DTYPE), exp


def compare(case, got, exp):
    name, U, Ch = case["name"], case["maxUnits"], case["maxChannels"]
    assert np.array_equal(got["results"]["status"], exp["results"]["status"]), (name, got["results"]["status"], exp["results"]["status"])
    ok = np.nonzero(exp["results"]["status"] == 0)[0]
    assert len(ok)
    for field in ("n_units", "n_channels", "flags", "bits_used"):
        assert np.array_equal(got["results"][field][ok], exp["results"][field][ok]), (name, field)
    for f in ok:
        n = int(exp["results"]["n_units"][f])
        assert got["units"][f * U:f * U + n].tobytes() == exp["units"][f * U:f * U + n].tobytes(), (name, "units of frame %d" % f)
        c = int(exp["results"]["n_channels"][f])
        blocks = slice(f * Ch, f * Ch + c)
        assert np.array_equal(got["q"][blocks], exp["q"][blocks]), (name, "spectrum of frame %d" % f)
        assert np.array_equal(got["meta"][blocks], exp["meta"][blocks]), (name, "band words of frame %d" % f)
        if case["wantTns"]:
            assert got["tns"][blocks].tobytes() == exp["tns"][blocks].tobytes(), (name, "TNS records of frame %d" % f)


def run_emulated(d):
    entries, counts = codebooks(d)
    emu = emu_lib.Emu()
    cases = json.load(open(os.path.join(d, "manifest.json")))
    for case in cases:
        data, frames, exp = load_case(d, case)
        got = emu_lib.emu_parse(emu, case["sampleIndex"], entries, counts, data, frames, case["maxUnits"], case["maxChannels"],
                                case["options"], case["wantTns"])
        compare(case, got, exp)
    return len(cases)


def test_library_codebooks_are_the_shipped_ones(standard):
    """aacg_standard_codebooks() (C, aacg_codebook_data.inc) == aac_codebooks.json as the JavaScript side expands it."""
    js_entries, js_counts = codebooks(standard)
    entries, counts = aacgpu.standard_codebooks()
    assert np.array_equal(counts, js_counts) and len(entries) == len(js_entries) == 1362
    o = 0
    for n in counts:
        key = lambda a: sorted(x.tobytes() for x in a)
        assert key(entries[o:o + n]) == key(js_entries[o:o + n])
        o += int(n)


def test_emulated_kernel_standard_codebooks(standard):
    assert run_emulated(standard) >= 10


def test_emulated_kernel_synthetic_codebooks(synthetic):
    assert run_emulated(synthetic) >= 10


def test_emulated_kernel_reads_in_place_when_lds_is_full(standard, monkeypatch):
    """Frames that do not fit the LDS staging arena are parsed from global memory: same results."""
    monkeypatch.setenv("AACG_EMU_ARENA", "2048")
    assert run_emulated(standard) >= 10


def test_kernel_source_under_address_sanitizer(standard):
    synthetic = standard
    """Garbage frames and every other case with exactly-sized buffers under ASan: no access outside what the ABI promises."""
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-C", emu_dir, "asan_parse"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for case in json.load(open(os.path.join(synthetic, "manifest.json"))):
        if case["name"] not in ("fuzz", "malformed", "five1_48", "mono44", "extras8k"):
            continue
        r = subprocess.run([os.path.join(emu_dir, "asan_parse"), synthetic, case["name"], str(case["sampleIndex"]), str(case["maxUnits"]),
                            str(case["maxChannels"]), str(case["options"]), "1" if case["wantTns"] else "0"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "asan_parse" in r.stdout, r.stdout + r.stderr[-2000:]


def adts_frame_table(data):
    """(offset, length) of every ADTS frame: 13-bit frame_length at bit 30 of the header."""
    out, off = [], 0
    while off + 7 <= len(data):
        assert data[off] == 0xFF and (data[off + 1] & 0xF0) == 0xF0
        length = ((int(data[off + 3]) & 3) << 11) | (int(data[off + 4]) << 3) | (int(data[off + 5]) >> 5)
        out.append((off, length))
        off += length
    return np.array(out, aacgpu.PARSE_FRAME_DTYPE)


STREAMS = os.path.join(ROOT, "tests", "golden", "streams")


def reference_stream(case):
    f = lambda ext, dt: np.fromfile(os.path.join(STREAMS, case["name"] + ext), dt)
    data, want_units = f(".aac", np.uint8), f(".units", aacgpu.UNIT_DTYPE)
    table = adts_frame_table(data)
    assert len(table) == case["frames"]
    return data, table, want_units, f(".q", np.int16).reshape(-1, 1024), f(".meta", np.uint16).reshape(-1, 120), f(".refpcm", np.float32)


def check_reference_stream_parse(case, got, want_units, want_q, want_meta, U):
    C = case["channels"]
    assert not got["results"]["status"].any(), case["name"]
    assert (got["results"]["n_units"] == U).all() and (got["results"]["n_channels"] == C).all()
    assert np.array_equal(got["q"], want_q), case["name"]
    assert np.array_equal(got["meta"], want_meta), case["name"]
    for field in ("n_ch", "flags", "channel"):
        assert np.array_equal(got["units"][field], want_units[field]), (case["name"], field)
    for field in ("window_sequence", "window_shape", "max_sfb", "group_count", "group_len"):
        assert np.array_equal(got["units"]["ch"][field], want_units["ch"][field]), (case["name"], field)


def test_emulated_kernel_on_the_reference_streams():
    """tests/golden/streams/*.aac were decoded by the reference itself (their .refpcm), and their .units / .q / .meta were
    checked field by field against the reference's own parse when they were generated (tests/js/test_frontend.js).
    The device parser's source, given the library's codebooks, reproduces them from the bytes."""
    entries, counts = aacgpu.standard_codebooks()
    emu = emu_lib.Emu()
    for case in json.load(open(os.path.join(STREAMS, "manifest.json"))):
        data, table, want_units, want_q, want_meta, _ = reference_stream(case)
        C, U = case["channels"], len(want_units) // case["frames"]
        got = emu_lib.emu_parse(emu, case["sampleIndex"], entries, counts, data, table, U, C, aacgpu.PARSE_REFERENCE_QUIRKS, False)
        check_reference_stream_parse(case, got, want_units, want_q, want_meta, U)


@pytest.mark.gpu
@pytest.mark.parametrize("case", json.load(open(os.path.join(STREAMS, "manifest.json"))), ids=lambda c: c["name"])
def test_gpu_reference_streams_from_bytes(case):
    """Bytes in, PCM out, everything on the device with the library's own codebooks: the committed .aac streams are
    parsed by aacg_parse_batch (equal to what the reference parsed, bit for bit), decoded by the engine, and the PCM is
    the one the reference's readChunk() produced from the same bytes (.refpcm) within 1e-5 RMS / 5e-6 of the signal."""
    data, table, want_units, want_q, want_meta, refpcm = reference_stream(case)
    C, U, n, si = case["channels"], len(want_units) // case["frames"], case["frames"], case["sampleIndex"]
    p = aacgpu.Parser(sample_index=si)                          # aacg_standard_codebooks()
    got = p.parse_batch(data, table, U, C, aacgpu.PARSE_REFERENCE_QUIRKS, False)
    check_reference_stream_parse(case, got, want_units, want_q, want_meta, U)
    units = got["units"]
    units["stream"] = 0
    units["n_out_ch"] = C
    units["pcm_offset"] = np.repeat(np.arange(n, dtype=np.uint32) * (1024 * C), U)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=1, max_channels=C, sample_index=si)
    pcm = eng.decode_batch(units, got["q"], got["meta"], n * 1024 * C)
    assert np.isfinite(pcm).all()
    d = pcm.astype(np.float64) - refpcm
    err, sig = float(np.sqrt(np.mean(d * d))), float(np.sqrt(np.mean(refpcm.astype(np.float64) ** 2)))
    assert sig > 1e-3 and err < 1e-5 and err <= 5e-6 * sig, (err, sig)
    eng.close()
    p.close()


def test_table_builder_refuses_bad_codebooks(synthetic):
    entries, counts = codebooks(synthetic)
    emu = emu_lib.Emu()
    data, frames = np.zeros(16, np.uint8), np.zeros(1, aacgpu.PARSE_FRAME_DTYPE)
    bad = entries.copy()
    bad["len"][5] += 1                                   # no longer complete
    with pytest.raises(RuntimeError, match="prefix code"):
        emu_lib.emu_parse(emu, 3, bad, counts, data, frames, 1, 1, 0, False)
    short = counts.copy()
    short[3] -= 1
    with pytest.raises(RuntimeError, match="entries"):
        emu_lib.emu_parse(emu, 3, entries, short, data, frames, 1, 1, 0, False)


@pytest.mark.gpu
def test_gpu_parse_matches_javascript_front_end(standard):
    synthetic = standard
    entries, counts = aacgpu.standard_codebooks()
    maps_before = "libaacgpu.so" in open("/proc/self/maps").read()
    for case in json.load(open(os.path.join(synthetic, "manifest.json"))):
        data, frames, exp = load_case(synthetic, case)
        p = aacgpu.Parser(entries, counts, sample_index=case["sampleIndex"])
        got = p.parse_batch(data, frames, case["maxUnits"], case["maxChannels"], case["options"], case["wantTns"])
        compare(case, got, exp)
        bad = np.nonzero(exp["results"]["status"])[0]
        for f in bad:
            assert p.status_string(got["results"]["status"][f]) != "unknown status"
        p.close()
    assert maps_before or "libaacgpu.so" in open("/proc/self/maps").read()


@pytest.mark.gpu
def test_gpu_bytes_to_pcm(standard, oracle):
    """Frames in, PCM out, both stages on the device: parse -> decode equals the oracle on the JavaScript front end's output."""
    synthetic = standard
    case = [c for c in json.load(open(os.path.join(synthetic, "manifest.json"))) if c["name"] == "stereo600"][0]
    data, frames, exp = load_case(synthetic, case)
    p = aacgpu.Parser(sample_index=3)
    got = p.parse_batch(data, frames, 1, 2, aacgpu.PARSE_REFERENCE_QUIRKS, False)
    assert not got["results"]["status"].any()
    n = len(frames)
    for units in (got["units"], exp["units"]):
        units["stream"] = 0
        units["n_out_ch"] = 2
        units["pcm_offset"] = np.arange(n, dtype=np.uint32) * 2048
        units["tns_offset"] = 0
        units["ch"]["flags"] = 0
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=1, max_channels=2)
    pcm = eng.decode_batch(got["units"], got["q"], got["meta"], n * 2048)
    ov = np.zeros((1, 2, 1024), np.float32)
    ref = oracle.decode_batch(exp["units"], exp["q"], exp["meta"], n * 2048, ov)
    d = pcm.astype(np.float64) - ref
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert sig > 1e-3 and float(np.sqrt(np.mean(d * d))) <= 5e-6 * sig
    eng.close()
    p.close()


@pytest.mark.gpu
def test_gpu_parser_edge_cases():
    """Empty batch, empty frame, a frame outside the buffer, a one-byte buffer, more elements than allowed for."""
    import aacgpu_workload
    entries, counts = aacgpu_workload.standin_codebooks()
    data, want_q, want_word = aacgpu_workload.tiny_frame(entries, counts)
    p = aacgpu.Parser(entries, counts, sample_index=3)
    none = p.parse_batch(data, np.zeros(0, aacgpu.PARSE_FRAME_DTYPE), 1, 1)
    assert len(none["results"]) == 0
    frames = np.zeros(4, aacgpu.PARSE_FRAME_DTYPE)
    frames["byte_offset"] = [0, 3, 0, len(data) - 1]
    frames["byte_length"] = [len(data), 0, 2, 1]                 # whole frame, empty, truncated twice
    out = p.parse_batch(data, frames, 1, 1)
    assert list(out["results"]["status"]) == [0, 1, 1, 1]
    assert np.array_equal(out["q"][0][:len(want_q)], want_q) and int(out["meta"][0][0]) == want_word
    assert not out["q"][1:].any()
    one = p.parse_batch(np.array([0xff], np.uint8), np.array([(0, 1)], aacgpu.PARSE_FRAME_DTYPE), 1, 1)     # END straight away: a frame without elements
    assert int(one["results"]["status"][0]) == 0 and int(one["results"]["n_units"][0]) == 0 and int(one["results"]["bits_used"][0]) == 8
    outside = np.array([(len(data) - 2, 8)], aacgpu.PARSE_FRAME_DTYPE)
    with pytest.raises(aacgpu.AacgError) as ei:
        p.parse_batch(data, outside, 1, 1)
    assert ei.value.code == -1
    twice = np.concatenate([data, np.zeros(1, np.uint8), data])  # the second copy at an odd offset
    frames2 = np.array([(0, len(data)), (len(data) + 1, len(data))], aacgpu.PARSE_FRAME_DTYPE)
    both = p.parse_batch(twice, frames2, 1, 1)
    assert not both["results"]["status"].any() and np.array_equal(both["q"][0], both["q"][1]) and np.array_equal(both["q"][1][:len(want_q)], want_q)
    p.close()
    bad = entries.copy()
    bad["len"][200] += 1
    with pytest.raises(aacgpu.AacgError) as ei:
        aacgpu.Parser(bad, counts, sample_index=3)
    assert ei.value.code == -1 and "prefix code" in str(ei.value)


def _silent(units):
    units["flags"] = 0
    units["ch"] = np.zeros((), units.dtype["ch"])
    units["ch"]["group_count"] = 1
    units["ch"]["group_len"][..., 0] = 1


@pytest.mark.gpu
@pytest.mark.parametrize("name,layout,moved", [("stereo600", [(2, 0)], False), ("fuzz", [(1, 0), (2, 1)], False), ("stereo600", [(2, 0)], True)])
def test_gpu_plan_refresh_from_parse(standard, oracle, name, layout, moved):
    """Parser -> transform without the host in between: a plan built once from the streams' structure, its device unit
    records rewritten from aacg_parse_device's output (aacg_plan_refresh_from_parse), equals parsing to the host and
    planning there — including the frames that become silent (refused by the parser, wrong element, noise bands)."""
    import torch
    synthetic = standard
    entries, counts = aacgpu.standard_codebooks()
    case = [c for c in json.load(open(os.path.join(synthetic, "manifest.json"))) if c["name"] == name][0]
    data, frames, _ = load_case(synthetic, case)
    U, Ch, n = case["maxUnits"], case["maxChannels"], len(frames)
    assert U == len(layout)
    C = sum(k for k, _ in layout)
    F = 10                                                     # frames per stream
    n = n // F * F
    frames = frames[:n]
    S = n // F
    p = aacgpu.Parser(entries, counts, sample_index=case["sampleIndex"])
    host = p.parse_batch(data, frames, U, Ch, case["options"], False)
    # the plan's skeleton: structure only
    skel = np.zeros(n * U, aacgpu.UNIT_DTYPE)
    _silent(skel)
    f = np.repeat(np.arange(n, dtype=np.uint32), U)
    skel["stream"] = f // F
    skel["pcm_offset"] = (f % F) * (1024 * C) + (f // F) * (F * 1024 * C)
    skel["n_out_ch"] = C
    skel["n_ch"] = np.tile(np.array([k for k, _ in layout], np.uint8), n)
    skel["channel"] = np.tile(np.array([c for _, c in layout], np.uint16), n)
    skel["coef_offset"] = skel["meta_offset"] = f * Ch + skel["channel"]
    if moved:
        # the plan's run tables carry copies of the block offsets (aacg_run.wave_coef): a frame the plan expects in OTHER blocks
        # than the parser's (frame * max_channels + channel) is not the frame the plan was made for — refused, silent, counted
        sel = (f % 7) == 3
        skel["coef_offset"][sel] = skel["meta_offset"][sel] = n * Ch + (f[sel] % 5) * Ch
    # host reference: the parsed records where they fit the skeleton, silent units elsewhere (the rule of the refresh kernel)
    ref_units = skel.copy()
    got = host["units"]
    res = np.repeat(host["results"], U)
    e = np.tile(np.arange(U), n)
    ok = (res["status"] == 0) & (e < res["n_units"]) & (got["n_ch"] == skel["n_ch"]) & (got["channel"] == skel["channel"]) & ((got["flags"] & 4) == 0)
    ok &= (got["coef_offset"] == skel["coef_offset"]) & (got["meta_offset"] == skel["meta_offset"])
    ref_units["flags"][ok] = got["flags"][ok]
    ch = got["ch"][ok].copy()
    ch["flags"] = 0
    ref_units["ch"][ok] = ch
    ref_units["coef_offset"] = got["coef_offset"]
    ref_units["meta_offset"] = got["meta_offset"]
    bad = ~ok
    qh, mh = host["q"], host["meta"]
    if moved:                                                  # (the oracle reads the silent units' blocks too: give it the extra ones)
        qh = np.concatenate([qh, np.zeros((5 * Ch, 1024), qh.dtype)])
        mh = np.concatenate([mh, np.zeros((5 * Ch,) + mh.shape[1:], mh.dtype)])
    ref_units["coef_offset"][bad] = skel["coef_offset"][bad]     # a refused frame keeps the planner's offsets: the silent unit still loads its blocks
    ref_units["meta_offset"][bad] = skel["meta_offset"][bad]
    ov = np.zeros((S, C, 1024), np.float32)
    ref = oracle.decode_batch(ref_units, qh, mh, n * 1024 * C, ov, sample_index=case["sampleIndex"])
    # device path
    dev = torch.device("cuda:0")
    t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(dev)
    pad = np.concatenate([data, np.zeros((-len(data)) % 16 + 32, np.uint8)])
    d_bytes, d_frames = t(pad), t(frames)
    d_units = torch.full((n * U * 64,), 0xFF, dtype=torch.uint8, device=dev)      # stale memory: a refused frame's record must not be trusted
    d_q = torch.zeros((n + 5) * Ch * 1024, dtype=torch.int16, device=dev)
    d_meta = torch.zeros((n + 5) * Ch * 120, dtype=torch.int16, device=dev)
    d_res = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
    d_pcm = torch.zeros(n * 1024 * C, dtype=torch.float32, device=dev)
    d_refused = torch.zeros(1, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=S, max_channels=C, sample_index=case["sampleIndex"])
    plan = eng.plan(skel)
    d_q.fill_(12345)                                          # with SKIP_ZERO_FILL the spectrum outside the coded bands is whatever was there
    torch.cuda.synchronize()
    p.parse_device(d_bytes.data_ptr(), d_frames.data_ptr(), n, U, Ch, case["options"] | aacgpu.PARSE_SKIP_ZERO_FILL, d_units.data_ptr(),
                   d_q.data_ptr(), d_meta.data_ptr(), None, d_res.data_ptr(), side.cuda_stream)
    eng.plan_refresh_from_parse(plan, d_units.data_ptr(), d_res.data_ptr(), U, d_refused.data_ptr(), side.cuda_stream)
    eng.decode_device(plan, d_q.data_ptr(), d_meta.data_ptr(), d_pcm.data_ptr(), side.cuda_stream)
    side.synchronize()
    assert int(d_refused.cpu()[0]) == int(bad.sum())
    if name == "stereo600" and not moved:
        assert not bad.any()
    else:
        assert bad.any() and ok.any()
    pcm = d_pcm.cpu().numpy()
    assert np.isfinite(pcm).all()
    d = pcm.astype(np.float64) - ref
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert sig > 1e-4 and float(np.sqrt(np.mean(d * d))) <= 5e-6 * sig + 1e-9
    plan.destroy()
    eng.close()
    p.close()
