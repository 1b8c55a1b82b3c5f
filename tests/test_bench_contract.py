"""The committed bench lines (profiles/r*_bench_*.json, written by bench.py on the GPU box) carry what the driver's contract and
the tier's measurement section ask for — a cheap guard against bench.py drifting away from it."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r03_bench_*.json")) +
               glob.glob(os.path.join(ROOT, "profiles", "r04_bench_*.json")))


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


@pytest.mark.parametrize("rnd", ["r02", "r03", "r04"])
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
        if "r03_" not in os.path.basename(path) and "r04_" not in os.path.basename(path):
            continue
        r = _load(path)["roofline"]
        assert r["copy_ceiling_GBs"] > 0 and abs(r["frac_of_copy"] - r["achieved"] / r["copy_ceiling_GBs"]) < 1e-9
        assert r["copy_ceiling_large_GBs"] > 1000 and r["kernel"].startswith("aacg_")


def test_round4_lines_report_the_median_of_repeated_regions():
    """Round 4: ms_per_step is the median of R >= 25 back-to-back repeats of the K-step region, with
    the spread beside it, and the process group's backend / world size are top-level facts of the line."""
    for path in LINES:
        if "r04_" not in os.path.basename(path):
            continue
        d = _load(path)
        t = d["timing"]
        assert t["repeats"] >= 25 and t["ms_per_step_min"] <= t["ms_per_step_median"] <= t["ms_per_step_max"]
        assert d["ms_per_step"] == t["ms_per_step_median"]
        assert d["collectives_backend_fallback"] is False and d["dist_world_size"] == d["n_gpus"]
        assert d["roofline"]["copy_timing"]["repeats"] >= 1


class _StubDist:
    """What bench.py asks of torch.distributed when it assembles its line."""
    def __init__(self, backend, world):
        self.backend, self.world = backend, world

    def get_backend(self):
        return self.backend

    def get_world_size(self):
        return self.world


def test_rank_local_arithmetic_of_the_config4_line_at_8_gpus():
    """`bench.py --gpus 8 --workload cfg4` without GPUs: everything a rank computes around its engine — the shape it decodes (32
    streams x 128 frames: its block of config 4's 256), its seed, the reduction of the repeats' times, the whole-job value, the
    fields that say which backend carried the barrier — with the engine's launch times stubbed."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
    import bench
    import aacgpu_shard
    world, K, R = 8, 20, 25
    mix, n_streams, n_frames, layout, n_chan = bench.workload_shape("cfg4")
    assert (mix, n_streams, n_frames, layout, n_chan) == (True, 32, 128, ("cpe",), 2)
    assert [aacgpu_shard.stream_shard(256, r, world) for r in range(world)] == [(32 * r, 32 * r + 32) for r in range(world)]
    seeds = [aacgpu_shard.rank_seed(0xAAC00002, r) for r in range(world)]
    assert len(set(seeds)) == world and seeds[0] == 0xAAC00002
    # stub engine: rank r's repeat takes K x (13.0 + 0.01 r) us, one slow outlier on rank 3; MAX over ranks, repeat by repeat
    per_rank = [[K * (0.0130 + 0.00001 * r) + (0.5 if (r == 3 and i == 7) else 0.0) for i in range(R)] for r in range(world)]
    region_ms = [max(per_rank[r][i] for r in range(world)) for i in range(R)]
    assert aacgpu_shard.reduce_max_list(None, per_rank[0]) == per_rank[0]
    st = bench.region_stats(region_ms, K)
    assert st["repeats"] == R and abs(st["ms_per_step_median"] - (0.0130 + 0.00007)) < 1e-12         # the outlier does not move the median
    assert st["ms_per_step_max"] > 0.03 and st["ms_per_step_min"] == st["ms_per_step_median"]
    value = bench.whole_job_value(world, n_streams * n_frames, st["ms_per_step_median"])
    assert abs(value - 8 * 4096 / (st["ms_per_step_median"] * 1e-3)) < 1e-6 * value
    assert bench.backend_fields("nccl", _StubDist("nccl", 8)) == {"dist_backend": "nccl", "dist_backend_requested": "nccl", "dist_world_size": 8,
                                                                  "collectives_backend_fallback": False}
    assert bench.backend_fields("nccl", _StubDist("gloo", 8))["collectives_backend_fallback"] is True
    assert bench.backend_fields("gloo", _StubDist("gloo", 2))["collectives_backend_fallback"] is False
    assert bench.backend_fields("nccl", None)["dist_world_size"] == 1
    # per-GPU algorithmic bytes of config 4: 128-frame chains
    assert bench.algorithmic_bytes_per_channel_frame("quant", n_frames) == 2048 + 240 + 4096 + 64
