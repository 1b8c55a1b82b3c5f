"""The JavaScript host (aac.js_amd/js) and its N-API addon, driven under Node like Aurora would."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"),
                                reason="node / node_api.h not present on this machine")


def build_addon():
    subprocess.run(["make", "-C", os.path.join(ROOT, "aac.js_amd", "napi")], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)


@needs_node
def test_host_cpu():
    """Unit packing matches the reference-derived records byte for byte; setCookie parses the ASC."""
    build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_host.js"), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host cpu tests ok" in r.stdout, r.stdout + r.stderr


@needs_node
def test_addon_fails_loudly_without_library():
    build_addon()
    code = ("const a=require(%r); try{a.load('/nonexistent/libaacgpu.so'); console.log('loaded')}"
            "catch(e){console.log('refused: '+e.message)}") % os.path.join(ROOT, "aac.js_amd", "napi", "aacgpu_napi.node")
    r = subprocess.run([NODE, "-e", code], capture_output=True, text=True, timeout=60)
    assert "refused" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@needs_node
def test_host_gpu():
    """decodeBatch through N-API and GpuAACDecoder.readChunk() with look-ahead, against the golden PCM."""
    build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_host.js"), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host gpu tests ok" in r.stdout, r.stdout + r.stderr


@needs_node
def test_js_port_matches_golden():
    """oracle/js/aac_port.js (the JavaScript CPU baseline) reproduces the reference's PCM on the ONLY_LONG frames."""
    import json
    r = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "aac_port.js"), "check", os.path.join(ROOT, "tests", "golden")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["rms"] < 1e-6


@needs_node
def test_aurora_registration_cpu():
    """aac.js_amd/js/aurora.js under an Aurora stand-in (tests/js/av_stub.js): 'mp4a' / 'aac ' registration, the ADTS demuxer's
    probe and events, and source -> demuxer -> decoder with a recording engine, however the source cuts the bytes."""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_aurora.js"), "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "aurora cpu tests ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@needs_node
def test_aurora_registration_gpu():
    """The same chain with the real engine: the PCM of Aurora's 'data' events == the reference's readChunk() output (.refpcm)."""
    build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_aurora.js"), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "aurora gpu tests ok" in r.stdout, r.stdout + r.stderr


@needs_node
def test_shared_engine_cpu():
    """aac.js_amd/js/shared_engine.js with recording engines: stream slots per sample rate, one batch for all decoders of a rate,
    every frame back to its own decoder in order, an engine error isolated to the stream that caused it, capacity, slot reuse."""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_shared.js"), "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shared cpu tests ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@needs_node
def test_shared_engine_gpu():
    """8 interleaved streams (5 sample rates / layouts) on one SharedEngine through the real engine: every stream equals the PCM
    the reference decoded from the same bytes (.refpcm) and, bit for bit, a decoder with an engine of its own."""
    build_addon()
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "test_shared.js"), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shared gpu tests ok" in r.stdout, r.stdout + r.stderr
