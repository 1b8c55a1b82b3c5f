"""Multi-GPU sharding harness on CPU (aac.js_amd/python/aacgpu_shard.py): world_size-2 gloo.

The path has no data-path collective (streams are sharded over ranks); the only collectives are the harness's
barrier, the MAX of the elapsed time and a SUM of counters.  Checked here, without a GPU: the shard arithmetic, the
launch line, that `self_launch` really starts N ranks as a child process and relays their exit code, and — with the
oracle standing in for a device (tests/shard_rank.py --decoder oracle) — the property that makes sharding valid: a
rank's streams decode identically alone or next to the others.  The same program runs the HIP engine per rank in the
-m gpu test (tests/test_gpu_parity.py::test_two_ranks_share_a_gpu)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import aacgpu_shard  # noqa: E402


def test_stream_shard_covers_every_stream_once():
    for total in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [aacgpu_shard.stream_shard(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert aacgpu_shard.stream_shard(256, 3, 8) == (96, 128)          # BASELINE config 4: streams [32r, 32r + 32)
    with pytest.raises(ValueError):
        aacgpu_shard.stream_shard(8, 2, 2)
    assert len({aacgpu_shard.rank_seed(0xAAC00002, r) for r in range(8)}) == 8


def test_launch_command_is_the_drivers_line():
    cmd = aacgpu_shard.launch_command(4, "bench.py", ["--gpus", "4", "--steps", "20"], port=29533, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "29533", "bench.py", "--gpus", "4", "--steps", "20"]
    assert not aacgpu_shard.launched_by_torchrun({})
    assert aacgpu_shard.launched_by_torchrun({"RANK": "0", "WORLD_SIZE": "2"})


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py --gpus 2 here (no GPU): it must start two ranks by itself and each must fail loudly — no CPU fallback."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stdout + r.stderr


def test_two_ranks_self_launched(tmp_path, oracle):
    out = str(tmp_path)
    S, T = 5, 6
    rc = aacgpu_shard.self_launch(2, os.path.join(ROOT, "tests", "shard_rank.py"),
                                  ["--out", out, "--streams", str(S), "--frames", str(T), "--decoder", "oracle"], timeout=600)
    assert rc == 0
    summary = json.load(open(os.path.join(out, "summary.json")))
    assert summary["world"] == 2 and summary["frames"] == S * T and summary["t_max"] >= summary["t_rank0"] > 0
    import aacgpu_workload
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, mix=True, intensity=True, seed=0xAAC00004)
    ov = np.zeros((S, 2, 1024), np.float32)
    whole = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
    parts = np.concatenate([np.fromfile(os.path.join(out, "pcm_rank%d.f32" % r), np.float32) for r in range(2)])
    assert parts.size == whole.size and np.array_equal(parts.view(np.uint32), whole.view(np.uint32))


def test_time_shard_partitions_the_frames():
    for n, world in ((128, 8), (17, 4), (3, 5), (1, 2)):
        spans = [aacgpu_shard.time_shard(n, r, world) for r in range(world)]
        assert [f for lo, hi, _ in spans for f in range(lo, hi)] == list(range(n))
        assert all(warm == (1 if lo > 0 and hi > lo else 0) for lo, hi, warm in spans)


def test_two_ranks_cut_the_streams_in_time(tmp_path, oracle):
    """SURVEY.md §8e, second partitioning: each rank takes a span of frames of every stream and recomputes the frame in
    front of its span for the tail; the spans, put together, are bit-identical to one decoder running straight through."""
    out = str(tmp_path)
    S, T = 3, 11
    rc = aacgpu_shard.self_launch(2, os.path.join(ROOT, "tests", "shard_rank.py"),
                                  ["--out", out, "--streams", str(S), "--frames", str(T), "--decoder", "oracle", "--shard", "time"], timeout=600)
    assert rc == 0
    summary = json.load(open(os.path.join(out, "summary.json")))
    assert summary["world"] == 2 and summary["frames"] == S * (T + 1)           # one warm-up frame per stream on rank 1
    import aacgpu_workload
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, mix=True, intensity=True, seed=0xAAC00004)
    ov = np.zeros((S, 2, 1024), np.float32)
    whole = oracle.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov).reshape(S, T, -1)
    spans = [aacgpu_shard.time_shard(T, r, 2) for r in range(2)]
    parts = [np.fromfile(os.path.join(out, "pcm_rank%d.f32" % r), np.float32).reshape(S, hi - lo, -1) for r, (lo, hi, _) in enumerate(spans)]
    got = np.concatenate(parts, axis=1)
    assert got.shape == whole.shape and np.array_equal(got.view(np.uint32), whole.view(np.uint32))


def test_self_launch_relays_the_exit_code(tmp_path):
    script = tmp_path / "fail.py"
    script.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert aacgpu_shard.self_launch(2, str(script), [], timeout=300) != 0
