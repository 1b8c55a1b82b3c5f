"""Sharding harness: one process per GPU, streams partitioned over ranks, no data-path collective.

The transform path has no arithmetic across streams (the only cross-frame state is the per-(stream,
channel) overlap buffer of src/filter_bank.js:38-41, which stays with the rank that owns the stream), so N GPUs are N
independent engines.  What this module holds is the part around them that bench.py, the tests and a host application
share:

  stream_shard(total, rank, world)   contiguous block of streams of a rank (BASELINE config 4: streams [32r, 32r+32))
  time_shard(frames, rank, world)    a single long stream cut in time: each rank recomputes one frame for its tail
  rank_seed(base, rank)              the synthetic generator's seed of a rank (independent data per rank, same shape)
  launch_command / self_launch       `--gpus N` without a launcher: start N ranks under torch.distributed.run as a CHILD
                                     process — before anything in this process has touched the GPU — and relay its output
  init_process_group / timed         rendezvous on 127.0.0.1; barrier + synchronize on both sides of a timed region, MAX
                                     of the elapsed time over ranks (the only collectives there are: a few bytes)

Collective backend: "nccl" (= RCCL over xGMI) when every rank has its own GPU; "gloo" for CPU tests and for ranks that
share one GPU (RCCL refuses two ranks on one device) — it only ever carries the barrier and 8-byte reductions.
"""
import os
import socket
import subprocess
import sys
import time


def stream_shard(total_streams, rank, world):
    """[lo, hi) of the streams rank owns: contiguous blocks, the remainder spread over the first ranks."""
    if not (0 <= rank < world) or total_streams < 0:
        raise ValueError("rank %d of %d" % (rank, world))
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def time_shard(n_frames, rank, world):
    """One long stream cut in time: rank r produces frames [lo, hi) of the stream and, when lo > 0, decodes
    frame lo - 1 in front of them for nothing but its windowed tail — the overlap state frame lo starts from depends on
    frame lo - 1's spectrum alone (src/filter_bank.js:109-118: `overlap[i] = buf[length + i] * window[...]`), so nothing
    has to come from the rank that owns the earlier frames.  Returns (lo, hi, warm): decode frames [lo - warm, hi) on a
    freshly reset stream and drop the PCM of the first `warm` (0 or 1) frames; cost 1 / (hi - lo) extra work."""
    lo, hi = stream_shard(n_frames, rank, world)
    return lo, hi, (1 if lo > 0 and hi > lo else 0)


def rank_seed(base_seed, rank):
    return (base_seed + 1000 * rank) & 0xFFFFFFFF


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_by_torchrun(env=None):
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def launch_command(n, script, script_args, port=None, python=None):
    """argv of the driver's own launch line (README / prompt contract), for a process that was started without it."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(script_args)


def self_launch(n, script, script_args, env=None, timeout=None):
    """Run `script` on n ranks as a child process group and relay its stdout / stderr.  Must be called before this
    process initialises the GPU (no torch.cuda call, no HIP call): the parent never becomes a rank, it only waits.
    Returns the child's exit code."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(launch_command(n, script, script_args), env=e)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()                                   # the exact child we started
        proc.wait()
        return 124


def init_process_group(backend, rank=None, world=None, device=None):
    """torch.distributed over 127.0.0.1; returns the module (or None for a single process without a launcher)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if rank is not None:
        kw.update(rank=rank, world_size=world)
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    try:
        dist.init_process_group(backend, **kw)
        if backend == "nccl":
            dist.barrier()                            # the communicator works, or we learn it here and not in the timed region
    except Exception as exc:                          # noqa: BLE001 - whatever RCCL raises
        if backend != "nccl":
            raise
        # RCCL unavailable between these ranks (IPC / topology): the harness needs a barrier and 8-byte reductions, which
        # gloo carries just as well; the data path has no collective either way.  Every rank sees the same failure.
        print("aacgpu_shard: RCCL unavailable (%s); the barrier falls back to gloo" % str(exc).splitlines()[0][:200], file=sys.stderr)
        if dist.is_initialized():
            dist.destroy_process_group()
        kw.pop("device_id", None)
        dist.init_process_group("gloo", **kw)
    return dist


def reduce_max(dist, value, device=None):
    """MAX over ranks of one float (the elapsed time of a timed region)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_max_list(dist, values, device=None):
    """Element-wise MAX over ranks of a list of floats (the R repeats of a timed region)."""
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def reduce_sum(dist, value, device=None):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def timed(dist, synchronize, body, device=None):
    """barrier + synchronize, body(), synchronize + barrier; returns (this rank's seconds, MAX over ranks)."""
    synchronize()
    if dist is not None:
        dist.barrier()
    synchronize()
    t0 = time.perf_counter()
    body()
    synchronize()
    if dist is not None:
        dist.barrier()
    synchronize()
    mine = time.perf_counter() - t0
    return mine, reduce_max(dist, mine, device)
