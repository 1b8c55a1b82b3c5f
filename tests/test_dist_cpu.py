"""Multi-GPU sharding logic on CPU: world_size-2 gloo.  The path has no data-path collective (streams are
sharded over ranks); the only collectives are the harness's barrier and MAX of elapsed time, exercised here
together with the property that makes sharding valid: a rank's streams decode identically alone or together."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
    import aacgpu_workload
    import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, T = 2, 5
    # rank r owns streams [S r, S r + S): same seeding rule as bench.py
    wl = aacgpu_workload.make_batch(n_streams=S, n_frames=T, mix=True, seed=0xAAC00002 + 1000 * rank)
    ov = np.zeros((S, 2, 1024), np.float32)
    pcm = orc.load().decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
    dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # bench.py's max-over-ranks
    frames = torch.tensor([wl["n_frames_total"]], dtype=torch.int64)
    dist.all_reduce(frames)                                          # whole-job count
    csum = torch.tensor([float(np.abs(pcm).sum())], dtype=torch.float64)
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, csum)
    if rank == 0:
        out.put((float(t.item()), int(frames.item()), [float(g.item()) for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_sharding():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, frames, sums = out.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert abs(tmax - 0.002) < 1e-12 and frames == 2 * 2 * 5
    # each rank's checksum equals decoding that rank's streams alone in this process (no cross-stream coupling)
    sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
    import aacgpu_workload
    import orc
    for r in range(2):
        wl = aacgpu_workload.make_batch(n_streams=2, n_frames=5, mix=True, seed=0xAAC00002 + 1000 * r)
        ov = np.zeros((2, 2, 1024), np.float32)
        pcm = orc.load().decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov)
        assert abs(float(np.abs(pcm).sum()) - sums[r]) < 1e-6 * max(1.0, sums[r])
