#!/usr/bin/env python3
"""Per-wave phase timeline of the run kernel (profiling aid; needs AACG_ABLATE including bit 16).
Usage: AACG_ABLATE=16 python tools/timeline.py [quant|spec]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import numpy as np, torch, aacgpu, aacgpu_workload

kind = sys.argv[1] if len(sys.argv) > 1 else "quant"
S, T = 256, int(os.environ.get("TL_FRAMES", "16"))
layout = tuple(os.environ.get("TL_LAYOUT", "cpe").split(","))          # e.g. TL_LAYOUT=sce: a folded mono chain per workgroup
wl = aacgpu_workload.make_batch(S, T, layout=layout, mix=bool(os.environ.get("TL_MIX")))
eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16 if kind == "quant" else aacgpu.INPUT_SPEC_F32, S, max(2, wl["C"]))
plan = eng.plan(wl["units"])
if kind == "quant":
    d_in = torch.from_numpy(wl["q"]).cuda(); d_meta = torch.from_numpy(wl["meta"].view(np.int16)).cuda(); mp = d_meta.data_ptr()
else:
    d_in = torch.randn(S * T * wl["C"], 1024, device="cuda") * 1000; mp = None
d_out = torch.empty(wl["n_pcm"], dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
PIPE = bool(os.environ.get("TL_PIPE"))                                  # TL_PIPE=1: launches through aacg_decode_pipelined (steady state: two in flight)
N_LAUNCH = int(os.environ.get("TL_LAUNCHES", "3001" if PIPE else "5"))    # pipelined: long enough for the host to be ahead of the GPU (steady state)
outs = [torch.empty(wl["n_pcm"], dtype=torch.float32, device="cuda") for _ in range(4)]
for i in range(N_LAUNCH):
    if PIPE:
        eng.decode_pipelined(plan, d_in.data_ptr(), mp, outs[i % 4].data_ptr())
    else:
        eng.decode_device(plan, d_in.data_ptr(), mp, d_out.data_ptr(), 0)
eng.synchronize()
raw = np.zeros(1 << 20, np.float32)
eng._check(eng.lib.aacg_get_table(eng.handle, 100, raw.ctypes.data, raw.size))
if PIPE:                                                                # launch n stamps quarter n mod 4 of the buffer: the last four launches are there
    # "this" launch = the fourth from the end, so that it has successors in flight like any launch of a long sequence; "previous" = the one before it
    parts = [raw[k << 18:][: 1 << 18] for k in range(4)]
    parts.sort(key=lambda a: float(np.median(a.view(np.uint64)[: 256 * 16 * 8].reshape(256, 16, 8)[:, 0, 0])))     # oldest first
    last, prev = parts[1], parts[0]
    tp = prev.view(np.uint64)[: 256 * 16 * 8].reshape(256, 16, 8).astype(np.float64) * 0.01
    raw = last
NW = int(os.environ.get("TL_WAVES", str(min(16, T))))                  # waves of a workgroup that carry a frame (T / 2 for a folded chain)
NB = int(os.environ.get("TL_BLOCKS", "256"))
t = raw.view(np.uint64)[: NB * 16 * 8].reshape(NB, 16, 8)[:, :NW].astype(np.float64) * 0.01     # 100 MHz ticks -> us
t0 = t[:, :, 0].min()
names = ["start", "tables+barrier", "spectrum staged", "imdct done", "prev tail seen", "stores issued", "loads landed"]
print("phase (us since first wave start): median over workgroups, by wave")
for w in range(NW):
    row = [np.median(t[:, w, k] - t0) for k in range(7)]
    print("wave %2d: " % w + "  ".join("%s %6.2f" % (names[k][:14], row[k]) for k in (0, 1, 6, 2, 3, 4, 5)))
print("kernel span (last stores issued - first start): %.2f us; start skew across WGs: %.2f us" % ((t[:, :, 5].max() - t0), t[:, :, 0].max() - t0))
if PIPE:
    NBp = NB
    sp, ep = tp[:NBp, :NW, 0].min(axis=1), tp[:NBp, :NW, 5].max(axis=1)           # previous launch: per-workgroup start, last stores issued
    sl, el = t[:, :, 0].min(axis=1), t[:, :, 5].max(axis=1)
    print("pipelined: previous launch's workgroups start %.2f .. %.2f, issue their last stores %.2f .. %.2f (us, same origin)" % (sp.min() - t0, sp.max() - t0, ep.min() - t0, ep.max() - t0))
    print("pipelined: this launch's workgroup starts, sorted, against the previous launch's workgroup ends (stores issued), sorted: median lag %.2f us, p10 %.2f, p90 %.2f" % tuple(
        np.percentile(np.sort(sl) - np.sort(ep), q) for q in (50, 10, 90)))
    print("pipelined: start-to-start of the two launches (median workgroup): %.2f us; workgroup life (start -> last stores issued) median %.2f us" % (np.median(sl) - np.median(sp), np.median(el - sl)))
    # the same CU, workgroup after workgroup: slot 7 = (CU id << 52) | clock when the wave's stores were acknowledged
    rawp, rawl = prev.view(np.uint64)[: 256 * 16 * 8].reshape(256, 16, 8), last.view(np.uint64)[: 256 * 16 * 8].reshape(256, 16, 8)
    M = (1 << 52) - 1
    cu_p, cu_l = (rawp[:NBp, 0, 7] >> np.uint64(52)).astype(int), (rawl[:NB, 0, 7] >> np.uint64(52)).astype(int)
    ack_p = ((rawp[:NBp, :NW, 7] & np.uint64(M)).astype(np.float64) * 0.01).max(axis=1)
    ack_l = ((rawl[:NB, :NW, 7] & np.uint64(M)).astype(np.float64) * 0.01).max(axis=1)
    by_cu = {int(c): i for i, c in enumerate(cu_p)}
    rows = [(ep[by_cu[c]], ack_p[by_cu[c]], sl[i]) for i, c in enumerate(cu_l) if int(c) in by_cu]
    if rows and len(set(cu_p.tolist())) > 1:
        r = np.array(rows)
        ok = r[:, 2] > r[:, 0]                          # this launch's workgroup came after the previous launch's on that CU
        r = r[ok]
        print("pipelined, CU by CU (%d CUs seen, %d pairs where this launch's workgroup followed the previous launch's on its CU):" % (len(set(cu_p.tolist())), len(r)))
        print("   last stores issued -> acknowledged: median %.2f us (p10 %.2f, p90 %.2f);  acknowledged -> next workgroup's first wave: median %.2f (p10 %.2f, p90 %.2f);  together %.2f" % (
            *(np.percentile(r[:, 1] - r[:, 0], q) for q in (50, 10, 90)), *(np.percentile(r[:, 2] - r[:, 1], q) for q in (50, 10, 90)), np.median(r[:, 2] - r[:, 0])))
    print("this launch: last stores issued -> acknowledged, per workgroup: median %.2f us, p90 %.2f" % tuple(np.percentile(ack_l - el, q) for q in (50, 90)))

# distribution over workgroups: where do the stragglers come from?
end = t[:, :, 5].max(axis=1) - t0                     # last stores issued per workgroup
start = t[:, :, 0].min(axis=1) - t0
print("per-workgroup end (last stores issued): " + "  ".join("p%d %.2f" % (q, np.percentile(end, q)) for q in (0, 10, 50, 90, 99, 100)))
print("per-workgroup duration: " + "  ".join("p%d %.2f" % (q, np.percentile(end - start, q)) for q in (0, 10, 50, 90, 99, 100)))
for x in range(8):
    sel = np.arange(NB) % 8 == x                     # blockIdx -> XCD round-robin
    print("XCD %d: start %.2f..%.2f  end median %.2f max %.2f" % (x, start[sel].min(), start[sel].max(), np.median(end[sel]), end[sel].max()))
late = np.argsort(end)[-8:]
print("latest workgroups:", [(int(i), round(float(start[i]), 2), round(float(end[i]), 2)) for i in late])
for k, nm in enumerate(names):
    if k in (4,): continue
    v = t[:, :, k] - t0
    print("%-16s all waves: p50 %.2f p90 %.2f p99 %.2f max %.2f" % (nm, *(np.percentile(v, q) for q in (50, 90, 99)), v.max()))
