"""The front-end corpus: 270 short AAC-LC streams (tests/js/corpus_cases.js regenerates them byte for byte wherever node runs) and
what the REFERENCE made of them (tests/golden/corpus.json, written by tests/golden/gen/gen_corpus.js from /root/reference/src in
the build container): SHA-256 of the quantised spectra, band words and unit records it parsed, checksum and 64 probe samples of
the PCM its readChunk() returned (src/decoder.js:125-216), and the message of the error it threw on a malformed frame
(src/ics.js:65-198, src/decoder.js:183-196, src/cpe.js:66).

not gpu: the device parser's SOURCE in the lane emulator reproduces the hashes from the bytes, and the oracle decodes the parsed
         records to the reference's PCM — 1 020 frames, all twelve sample rates, one to seven channels.
gpu:     aacg_parse_batch + the engine, and the resident route (aacg_pipeline_*), do the same on the device; a malformed frame
         yields the status whose string is the reference's message.
"""
import base64
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import aacgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORPUS = json.load(open(os.path.join(ROOT, "tests", "golden", "corpus.json")))["streams"]


def sha(buf):
    return base64.b64encode(hashlib.sha256(bytes(buf)).digest()).decode().rstrip("=")


@pytest.fixture(scope="module")
def streams(tmp_path_factory):
    """The corpus regenerated here and now; every stream's bytes are the ones the reference was given."""
    d = str(tmp_path_factory.mktemp("corpus"))
    r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "corpus_cases.js"), d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {}
    for e in CORPUS:
        data = np.fromfile(os.path.join(d, e["name"] + ".aac"), np.uint8)
        assert sha(data)[:16] == e["bytes"], e["name"] + ": the regenerated stream is not the one the reference decoded"
        out[e["name"]] = data
    return out


def frame_table(data):
    out, off = [], 0
    while off + 7 <= len(data):
        assert data[off] == 0xFF and (data[off + 1] & 0xF0) == 0xF0
        length = ((int(data[off + 3]) & 3) << 11) | (int(data[off + 4]) << 3) | (int(data[off + 5]) >> 5)
        out.append((off, length))
        off += length
    return np.array(out, aacgpu.PARSE_FRAME_DTYPE)


def canonical_units(units):
    """gen_corpus.js canonicalUnits: what the front end and the device parser both produce, without the caller's fields."""
    out = bytearray()
    for u in units:
        n = int(u["n_ch"])
        out += bytes([n, int(u["flags"]), int(u["channel"]) & 255, int(u["channel"]) >> 8])
        for c in range(n):
            ch = u["ch"][c]
            out += bytes([int(ch["window_sequence"]), int(ch["window_shape"]), int(ch["max_sfb"]), int(ch["group_count"])]) + bytes(int(g) for g in ch["group_len"])
    return bytes(out)


def check_parse(e, got, U):
    """The parse of the frames the reference decoded equals the reference's (by hash); the frame it gave up on has the status
    whose text is its message."""
    good, C = e["decoded"], e["channels"]
    res = got["results"]
    assert not res["status"][:good].any(), (e["name"], res["status"])
    if e["error"]:
        t = e["error"]["frame"]
        text = aacgpu.load_library().aacg_parse_status_string(int(res["status"][t])).decode()
        assert res["status"][t] != 0 and e["error"]["message"].startswith(text), (e["name"], text, e["error"]["message"])
    if not good:
        return None
    n_el = e["n_units"] // good
    assert (res["n_units"][:good] == n_el).all() and (res["n_channels"][:good] == C).all(), e["name"]
    units = got["units"].reshape(-1, U)[:good, :n_el].reshape(-1)
    q = got["q"].reshape(-1, got["q"].shape[0] // len(res), 1024)[:good, :C].reshape(-1, 1024)
    meta = got["meta"].reshape(-1, got["meta"].shape[0] // len(res), 120)[:good, :C].reshape(-1, 120)
    assert sha(np.ascontiguousarray(q)) == e["q"], e["name"] + ": quantised spectra"
    assert sha(np.ascontiguousarray(meta)) == e["meta"], e["name"] + ": band words"
    assert sha(canonical_units(units)) == e["units"], e["name"] + ": unit records"
    return units.copy(), np.ascontiguousarray(q), np.ascontiguousarray(meta)


def check_pcm(e, pcm):
    p = e["pcm"]
    assert pcm.size == p["n"] and np.isfinite(pcm).all(), e["name"]
    probes = np.frombuffer(base64.b64decode(p["probes"]), np.float32)
    idx = [((k * 7919 + 13) * 104729) % p["n"] for k in range(64)]
    rms = (p["sumsq"] / p["n"]) ** 0.5
    assert np.abs(pcm[idx].astype(np.float64) - probes).max() <= 1e-5 * max(1.0, 4.0 * rms), (e["name"], float(np.abs(pcm[idx] - probes).max()), rms)
    x = pcm.astype(np.float64)
    assert abs(float(x.sum()) - p["sum"]) <= 2e-6 * p["n"] ** 0.5 * max(rms, 1e-3) + 1e-9 * p["n"], (e["name"], float(x.sum()), p["sum"])
    assert abs(float((x * x).sum()) - p["sumsq"]) <= 2e-5 * p["sumsq"] + 1e-12, (e["name"], float((x * x).sum()), p["sumsq"])


def prepared_units(units, n_frames, n_el, C):
    units["stream"] = 0
    units["n_out_ch"] = C
    units["pcm_offset"] = np.repeat(np.arange(n_frames, dtype=np.uint32) * (1024 * C), n_el)
    units["coef_offset"] = units["meta_offset"] = (np.repeat(np.arange(n_frames, dtype=np.uint32) * C, n_el) + units["channel"]).astype(np.uint32)
    return units


def test_corpus_covers_what_it_claims():
    names = [e["name"] for e in CORPUS]
    assert len(CORPUS) >= 200 and len(set(names)) == len(names)
    assert {e["si"] for e in CORPUS} == set(range(12))
    assert {e["channels"] for e in CORPUS} >= {1, 2, 3, 6, 7}
    msgs = {e["error"]["message"] for e in CORPUS if e["error"]}
    assert len(msgs) >= 9 and sum(e["decoded"] for e in CORPUS) >= 1000
    assert os.path.getsize(os.path.join(ROOT, "tests", "golden", "corpus.json")) < 200 * 1024


def test_emulated_parser_and_oracle_on_the_corpus(streams, oracle):
    """CPU: the device parser's source (lane emulator) on every stream — the reference's hashes and error messages — and the
    oracle on the parsed records — the reference's PCM."""
    import emu_lib
    entries, counts = aacgpu.standard_codebooks()
    emu = emu_lib.Emu()
    for e in CORPUS:
        data, C = streams[e["name"]], e["channels"]
        table = frame_table(data)
        assert len(table) == e["frames"]
        U = 8
        got = emu_lib.emu_parse(emu, e["si"], entries, counts, data, table, U, 8, aacgpu.PARSE_REFERENCE_QUIRKS, False)
        parsed = check_parse(e, got, U)
        if parsed is None:
            continue
        units, q, meta = parsed
        n_el = e["n_units"] // e["decoded"]
        units = prepared_units(units, e["decoded"], n_el, C)
        ov = np.zeros((1, C, 1024), np.float32)
        check_pcm(e, oracle.decode_batch(units, q, meta, e["decoded"] * 1024 * C, ov, sample_index=e["si"]))


@pytest.mark.gpu
def test_gpu_parser_and_engine_on_the_corpus(streams):
    """GPU: aacg_parse_batch on every stream (hashes, error statuses), the engine on what it parsed (PCM)."""
    parsers, engines = {}, {}
    for e in CORPUS:
        data, C, si = streams[e["name"]], e["channels"], e["si"]
        table = frame_table(data)
        p = parsers.get(si) or parsers.setdefault(si, aacgpu.Parser(sample_index=si))
        U = 8
        got = p.parse_batch(data, table, U, 8, aacgpu.PARSE_REFERENCE_QUIRKS, False)
        parsed = check_parse(e, got, U)
        if parsed is None:
            continue
        units, q, meta = parsed
        units = prepared_units(units, e["decoded"], e["n_units"] // e["decoded"], C)
        eng = engines.get((si, C)) or engines.setdefault((si, C), aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=1, max_channels=C, sample_index=si))
        eng.reset_stream(0)
        check_pcm(e, eng.decode_batch(units, q, meta, e["decoded"] * 1024 * C))
        # the same stream as MP4 samples: bare raw_data_blocks (the ADTS header cut off) parse to the same records
        if e["name"].startswith("c") and int(e["name"][1:4]) % 4 == 0:
            hdr = np.array([7 if data[int(o) + 1] & 1 else 9 for o in table["byte_offset"]], np.uint32)
            bare = table.copy()
            bare["byte_offset"] += hdr
            bare["byte_length"] -= hdr
            check_parse(e, p.parse_batch(data, bare, U, 8, aacgpu.PARSE_REFERENCE_QUIRKS, False), U)
    for x in list(parsers.values()) + list(engines.values()):
        x.close()


@pytest.mark.gpu
def test_gpu_resident_route_on_the_corpus(streams):
    """GPU: the clean streams through aacg_pipeline_* — one pipeline per (sample rate, channel count), every stream of the group
    a slot of its own, in batches of two frames with three of them in flight; the PCM is the reference's."""
    groups = {}
    for e in CORPUS:
        if not e["error"]:
            groups.setdefault((e["si"], e["channels"]), []).append(e)
    for (si, C), members in sorted(groups.items()):
        F = min(e["frames"] for e in members)
        S = len(members)
        pipe = aacgpu.Pipeline(channels=C, max_streams=S, max_frames=2, sample_index=si)
        tabs = [frame_table(streams[e["name"]]) for e in members]
        base = np.cumsum([0] + [len(streams[e["name"]]) for e in members]).astype(np.uint32)
        data = np.concatenate([streams[e["name"]] for e in members])
        out = [np.zeros(e["frames"] * 1024 * C, np.float32) for e in members]
        tickets = []
        for a in range(0, F, 2):
            n = min(2, F - a)
            fr = np.zeros(S * n, aacgpu.PARSE_FRAME_DTYPE)
            for s in range(S):
                fr[s * n:(s + 1) * n] = tabs[s][a:a + n]
                fr["byte_offset"][s * n:(s + 1) * n] += base[s]
            tickets.append((a, n, pipe.submit(data, fr, np.arange(S), n)))
        for a, n, t in tickets:
            pcm, res, refused = pipe.collect(t)
            assert refused == 0 and not res["status"].any(), (si, C, a)
            for s in range(S):
                out[s][a * 1024 * C:(a + n) * 1024 * C] = pcm.reshape(S, n * 1024 * C)[s]
        for s, e in enumerate(members):
            if e["frames"] == F:
                check_pcm(e, out[s])
            else:                                         # a longer stream: its first F frames' probes where they fall
                p = e["pcm"]
                probes = np.frombuffer(base64.b64decode(p["probes"]), np.float32)
                idx = np.array([((k * 7919 + 13) * 104729) % p["n"] for k in range(64)])
                inside = idx < F * 1024 * C
                rms = (p["sumsq"] / p["n"]) ** 0.5
                assert np.abs(out[s][idx[inside]] - probes[inside]).max() <= 1e-5 * max(1.0, 4.0 * rms), e["name"]
        pipe.close()
