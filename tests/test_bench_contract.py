"""The committed bench lines (profiles/r*_bench_*.json, written by bench.py on the GPU box) carry what the driver's contract and
the tier's measurement section ask for — a cheap guard against bench.py drifting away from it."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r03_bench_*.json")))


def _load(path):
    with open(path) as f:
        return json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = _load(path)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "parity_rms"):
        assert key in d, key
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # value is whole-job throughput: frames of all ranks per step / the measured step time
    frames = d["config"]["streams_per_gpu"] * d["config"]["frames_per_stream_per_step"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["output_ok"] is True and d["parity_rms"] < 1e-4                      # BASELINE's bound; the run's own gate is tighter
    assert d["parity"]["rms"] <= d["parity"]["gate_rms"] and d["parity"]["rel"] <= d["parity"]["gate_rel"]


@pytest.mark.parametrize("rnd", ["r02", "r03"])
def test_headline_line_has_the_cpu_baseline_and_traffic(rnd):
    d = _load(os.path.join(ROOT, "profiles", rnd + "_bench_quant.json"))
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] > 1 and cb["value"] > cb["single_core"]["value"] > 0 and cb["sample"]
    assert d["n_gpus"] == 1 and d["roofline"]["kernel"] == "aacg_imdct_run_quant"
    assert d["roofline"]["traffic"] and 0.9 < d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"] < 1.2
    assert "config 2" in d["config"]["workload"]


def test_round3_lines_carry_the_same_run_copy_ceiling():
    """Round 3: every line measures, in the same process right behind the timed region, what a float4 copy launch of the step's
    byte volume gets on that box (aacg_calib_copy) — boxes of the pool differ by several per cent, `frac_of_copy` does not."""
    for path in LINES:
        if "r03_" not in os.path.basename(path):
            continue
        r = _load(path)["roofline"]
        assert r["copy_ceiling_GBs"] > 0 and abs(r["frac_of_copy"] - r["achieved"] / r["copy_ceiling_GBs"]) < 1e-9
        assert r["copy_ceiling_large_GBs"] > 1000 and r["kernel"].startswith("aacg_")
