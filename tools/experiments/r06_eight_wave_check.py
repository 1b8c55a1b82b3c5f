#!/usr/bin/env python3
"""experiment: the 8-wave workgroups (AACG_DEBUG_ROUTE_HALF_RUNS) against launch behind launch, bit for bit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, aacgpu, aacgpu_workload
HALF = 16
def overlaps(eng, S, C): return np.stack([[eng.get_overlap(s, c) for c in range(C)] for s in range(S)])
for layout, S, T, n, seam, mix in [(("cpe",), 256, 16, 40, "q", False), (("cpe",), 256, 16, 40, "q", True), (("cpe",), 300, 5, 40, "q", True), (("cpe",), 32, 128, 40, "q", True), (("sce",), 700, 3, 40, "q", True), (("cpe",), 256, 16, 40, "f", True)]:
    base = aacgpu_workload.make_batch(n_streams=S, n_frames=T, layout=layout, mix=mix, intensity=mix, seed=7300)
    C = base["C"]
    kind = aacgpu.INPUT_QUANT_I16 if seam == "q" else aacgpu.INPUT_SPEC_F32
    rng = np.random.default_rng(7)
    ins = []
    for j in range(n):
        q = base["q"] if j == 0 else (np.roll(base["q"], 131 * j, axis=0) * rng.choice([-1, 1])).astype(np.int16)
        h = (np.sign(q) * np.abs(q.astype(np.float32)) ** (4.0 / 3.0) * 2.0 ** 4).astype(np.float32) if seam == "f" else np.ascontiguousarray(q)
        ins.append(torch.from_numpy(h).cuda())
    d_meta = torch.from_numpy(base["meta"].view(np.int16)).cuda() if seam == "q" else None
    mp = d_meta.data_ptr() if d_meta is not None else None
    results = []
    for mode in ("serial", "half"):
        eng = aacgpu.Engine(kind, max_streams=S, max_channels=C)
        if mode == "half": eng.debug_set_route(HALF)
        plan = eng.plan(base["units"])
        name = eng.plan_kernels(plan, pipelined=(mode == "half"))
        outs = [torch.full((base["n_pcm"],), float("nan"), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        for j in range(n):
            if mode == "half": eng.decode_pipelined(plan, ins[j].data_ptr(), mp, outs[j].data_ptr())
            else: eng.decode_device(plan, ins[j].data_ptr(), mp, outs[j].data_ptr(), 0)
        eng.synchronize(); torch.cuda.synchronize()
        results.append((name, [o.cpu().numpy() for o in outs], overlaps(eng, S, C)))
        plan.destroy(); eng.close()
    (n0, a, sa), (n1, b, sb) = results
    bad = [j for j in range(n) if not np.array_equal(a[j].view(np.uint32), b[j].view(np.uint32))]
    print(layout, S, T, seam, mix, n0, "|", n1, "| launches that differ:", bad[:8], "| nan:", any(np.isnan(x).any() for x in b), "| state equal:", np.array_equal(sa.view(np.uint32), sb.view(np.uint32)), flush=True)
    if bad:
        j = bad[0]; d = np.nonzero(a[j].view(np.uint32) != b[j].view(np.uint32))[0]
        print("   first differing launch", j, ":", len(d), "samples, first at", d[:6], "frames", sorted(set((d // (1024 * C)) % T))[:16])
