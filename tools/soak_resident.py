#!/usr/bin/env python3
"""Soak of the resident route (aacg_pipeline_*) on a GPU box (not a pytest): for a few minutes, random groups of the corpus's
streams (tests/js/corpus_cases.js: twelve sample rates, one to eight channels, every syntax element the path reads; the
malformed ones too) go through a pipeline in random batches — streams per batch, frames per stream, lanes, batches in flight,
int16 or f32 PCM all drawn per round; now and then a stream is reset and starts over.  Expected: what the ORACLE makes of the
records aacg_parse_batch parsed from the same bytes (the parser itself is pinned by the reference on the same corpus,
tests/test_corpus.py), frame by frame with the overlap state carried along; a malformed frame must come back with the status
whose text is the reference's message and must not disturb its neighbours in the batch.
Usage: python tools/soak_resident.py [seconds=120]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aacgpu  # noqa: E402
import orc  # noqa: E402
import test_corpus as T  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
o = orc.load()
rng = np.random.default_rng(20261003)
d = tempfile.mkdtemp()
r = subprocess.run(["node", os.path.join(ROOT, "tests", "js", "corpus_cases.js"), d], capture_output=True, text=True)
assert r.returncode == 0, r.stdout + r.stderr
lib = aacgpu.load_library()

# per stream: bytes, frame table, and per frame the oracle's PCM (None from the frame the reference gave up on)
groups, parsers = {}, {}
for e in T.CORPUS:
    data = np.fromfile(os.path.join(d, e["name"] + ".aac"), np.uint8)
    table = T.frame_table(data)
    C, si, good = e["channels"], e["si"], e["decoded"]
    p = parsers.get(si) or parsers.setdefault(si, aacgpu.Parser(sample_index=si))
    got = p.parse_batch(data, table, 8, 8, aacgpu.PARSE_REFERENCE_QUIRKS, False)
    ref = None
    if good:
        n_el = e["n_units"] // good
        units = got["units"].reshape(-1, 8)[:good, :n_el].reshape(-1).copy()
        q = np.ascontiguousarray(got["q"].reshape(len(table), -1, 1024)[:good, :C].reshape(-1, 1024))
        meta = np.ascontiguousarray(got["meta"].reshape(len(table), -1, 120)[:good, :C].reshape(-1, 120))
        units = T.prepared_units(units, good, n_el, C)
        ref = o.decode_batch(units, q, meta, good * 1024 * C, np.zeros((1, C, 1024), np.float32), sample_index=si).reshape(good, 1024 * C)
    groups.setdefault((si, C), []).append(dict(e=e, data=data, table=table, ref=ref, status=got["results"]["status"].copy()))
for p in parsers.values():
    p.close()

t0 = time.time()
rounds = batches = frames = refused_frames = resets = 0
worst = 0.0
keys = sorted(groups)
while time.time() - t0 < budget:
    si, C = keys[int(rng.integers(0, len(keys)))]
    members = groups[(si, C)]
    S = int(rng.integers(1, min(len(members), 24) + 1))
    chosen = [members[i] for i in rng.permutation(len(members))[:S]]
    # a malformed stream stays in only up to (and including) its bad frame: what comes after a frame the reference threw on is ours to define
    Fmax = int(rng.integers(1, 5))
    lanes = int(rng.integers(1, 7))
    i16 = bool(rng.integers(0, 3) == 0)
    pipe = aacgpu.Pipeline(channels=C, max_streams=S, max_frames=Fmax, sample_index=si, lanes=lanes,
                           output_kind=aacgpu.OUTPUT_I16 if i16 else aacgpu.OUTPUT_F32)
    base = np.cumsum([0] + [len(m["data"]) for m in chosen]).astype(np.uint32)
    data = np.concatenate([m["data"] for m in chosen])
    usable = [m["e"]["decoded"] + (1 if m["e"]["error"] else 0) for m in chosen]      # frames of each stream that go in
    pos = [0] * S                                   # next frame of each stream
    pending = []
    def submit():
        F = int(rng.integers(1, Fmax + 1))
        F = min([F] + [usable[s] - pos[s] for s in range(S)])
        if F <= 0:
            return False
        fr = np.zeros(S * F, aacgpu.PARSE_FRAME_DTYPE)
        for s in range(S):
            fr[s * F:(s + 1) * F] = chosen[s]["table"][pos[s]:pos[s] + F]
            fr["byte_offset"][s * F:(s + 1) * F] += base[s]
        pending.append((pipe.submit(data, fr, np.arange(S), F), F, list(pos)))
        for s in range(S):
            pos[s] += F
        return True
    def collect():
        global batches, frames, refused_frames, worst
        t, F, at = pending.pop(0)
        pcm, res, refused = pipe.collect(t)
        pcm = pcm.reshape(S, F, 1024 * C)
        n_bad = 0
        for s in range(S):
            m = chosen[s]
            for f in range(F):
                k = at[s] + f
                st = int(res["status"][s * F + f])
                if k >= m["e"]["decoded"]:                                  # the frame the reference threw on
                    text = lib.aacg_parse_status_string(st).decode()
                    assert st != 0 and m["e"]["error"]["message"].startswith(text), (m["e"]["name"], k, text)
                    assert np.isfinite(pcm[s, f].astype(np.float64)).all(), (m["e"]["name"], k)      # (its units are silent ones: the frame holds the tail of the one before it)
                    n_bad += 1
                    continue
                assert st == 0, (m["e"]["name"], k, st)
                want = m["ref"][k]
                got = pcm[s, f].astype(np.float64) / (32768.0 if i16 else 1.0)
                if i16:
                    want = np.clip(np.rint(want.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0
                rms = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
                err = float(np.abs(got - want).max())
                # int16: one step for the rounding, the float route's allowance, and 3e-6 of the frame's PEAK: the corpus has spiky frames
                # that peak at 30 with an rms under 1, and a transform's rounding error goes with its largest values (measured over the
                # corpus: f32 route <= 4.7e-6 of the peak, int16 route <= 2.5e-6 beyond its half step) — two steps for a sample in range
                peak = float(np.abs(m["ref"][k]).max())
                tol = 1e-5 * max(1.0, 4.0 * rms) + (1.01 / 32768.0 + 3e-6 * peak if i16 else 0.0)
                worst = max(worst, err / tol)
                assert err <= tol, (m["e"]["name"], k, err, tol, "i16" if i16 else "f32", lanes, S, F)
        assert refused == n_bad, (refused, n_bad, si, C, S, F, lanes, [(chosen[s]["e"]["name"], at[s]) for s in range(S)], res["status"].tolist(), res["n_units"].tolist())
        batches += 1
        frames += S * F
        refused_frames += n_bad
    alive = True
    while alive or pending:
        depth = int(rng.integers(0, lanes))
        while alive and len(pending) <= depth:
            alive = submit()
        if pending:
            collect()
        if alive and not pending and rng.random() < 0.05:                   # nothing in flight: a stream starts over
            s = int(rng.integers(0, S))
            pipe.reset_stream(s)
            pos[s] = 0                                                      # (a reset stream starts over at its frame 0: the oracle's frames apply again)
            resets += 1
    pipe.close()
    rounds += 1
print("soak_resident OK: %.0f s, %d pipelines, %d batches, %d frames (%d of them malformed: refused with the reference's message), %d resets; "
      "worst error %.2f of the tolerance (f32: 1e-5 x max(1, 4 rms) per sample; int16: that + one step + 3e-6 of the frame's peak)" % (time.time() - t0, rounds, batches, frames, refused_frames, resets, worst))
