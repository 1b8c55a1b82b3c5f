"""The JavaScript bitstream front end (aac.js_amd/js/frontend.js): ADTS / raw_data_block bytes -> engine input.

tests/js/test_frontend.js does the parsing checks under Node (writer -> FrontEnd round trips, and — in the build
container, where the reference checkout exists — the same bytes through the reference's own readChunk(), compared
field by field and integer by integer).  It also writes, per stream, the engine input FrontEnd + GpuAACDecoder
produced and the PCM the reference decoded from the same bytes; those files are committed under
tests/golden/streams/ (data: .aac bytes in, numbers out) so that the rest of the path can be checked where the
reference is absent:

  not gpu : the committed engine inputs through the oracle and the emulated kernels == the reference's PCM;
            in the build container, regenerating the files reproduces the committed ones byte for byte
  gpu     : the same through the HIP engine
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import emu_lib
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
NODE = shutil.which("node")
RMS_TOL = 1e-5          # BASELINE.json's bound is 1e-4 RMS on [-1,1) PCM
REL_TOL = 5e-6


def manifest():
    return json.load(open(os.path.join(STREAMS, "manifest.json")))


def load(case):
    name = case["name"]
    f = lambda ext, dt: np.fromfile(os.path.join(STREAMS, name + ext), dt)
    return (f(".units", np.uint8).view(orc.UNIT_DTYPE).ravel(), f(".q", np.int16), f(".meta", np.uint16), f(".refpcm", np.float32))


def check(pcm, ref):
    assert not np.isnan(pcm).any()
    d = pcm.astype(np.float64) - ref
    err, sig = float(np.sqrt(np.mean(d * d))), float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert sig > 1e-3
    assert err < RMS_TOL and err <= REL_TOL * sig, (err, sig)


@pytest.mark.skipif(NODE is None, reason="node not present")
def test_frontend_js(tmp_path):
    """Parser checks under Node; with the reference present, the regenerated stream files equal the committed ones."""
    out = str(tmp_path / "streams")
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_frontend.js"), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frontend tests passed" in r.stdout
    if os.path.exists("/root/reference/src/decoder.js"):
        names = sorted(os.listdir(STREAMS))
        assert names == sorted(os.listdir(out))
        for n in names:
            assert open(os.path.join(out, n), "rb").read() == open(os.path.join(STREAMS, n), "rb").read(), n


@pytest.mark.parametrize("case", manifest(), ids=lambda c: c["name"])
def test_streams_oracle_and_emulator(oracle, case):
    units, q, meta, ref = load(case)
    C, si = case["channels"], case["sampleIndex"]
    assert len(ref) == case["frames"] * C * 1024
    ov = np.zeros((1, C, 1024), np.float32)
    check(oracle.decode_batch(units, q, meta, ref.size, ov, sample_index=si), ref)
    pool = np.zeros((1, C, emu_lib.OV_BUFFERS, 1024), np.float32)
    par = np.zeros(C, np.uint8)
    check(emu_lib.Emu().decode(units, q, meta, ref.size, pool, par, sample_index=si), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", manifest(), ids=lambda c: c["name"])
def test_streams_gpu(case):
    import aacgpu
    units, q, meta, ref = load(case)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=1, max_channels=case["channels"], sample_index=case["sampleIndex"])
    check(eng.decode_batch(units, q, meta, ref.size), ref)
    eng.close()
