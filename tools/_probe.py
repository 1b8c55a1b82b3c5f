import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
import numpy as np, torch, aacgpu, aacgpu_workload
base = aacgpu_workload.make_batch(n_streams=256, n_frames=16, seed=1)
eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, max_streams=256, max_channels=2)
plan = eng.plan(base["units"])
d_meta = torch.from_numpy(base["meta"].view(np.int16)).cuda()
bufs = [(torch.from_numpy(np.roll(base["q"], 131 * b, axis=0).copy()).cuda(), torch.empty(base["n_pcm"], dtype=torch.float32, device="cuda")) for b in range(8)]
vp = ctypes.c_void_p
ptrs = [(vp(a.data_ptr()), vp(o.data_ptr())) for a, o in bufs]
h, ph, meta = vp(eng.handle.value), vp(plan.handle.value), vp(d_meta.data_ptr())
lib = eng.lib
def run(n, marks, every):
    t0 = time.perf_counter()
    for i in range(n):
        a, o = ptrs[i % 8]
        k = i % every
        if marks is not None and k >= every - 3:
            lib.aacg_decode_pipelined_timed(h, ph, a, meta, o, marks[(i // every) * 3 + (k - (every - 3))].h)
        else:
            lib.aacg_decode_pipelined(h, ph, a, meta, o)
    t1 = time.perf_counter()
    eng.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6
for _ in range(3): run(4000, None, 20)
print("no marks:                         enqueue %.2f us/launch, total %.2f" % run(20000, None, 20))
fresh = [aacgpu.TimerMark() for _ in range(3000)]
print("3 fresh marks per 20 launches:     enqueue %.2f us/launch, total %.2f" % run(20000, fresh, 20))
print("the same marks bound a second time: enqueue %.2f us/launch, total %.2f" % run(20000, fresh, 20))
fresh2 = [aacgpu.TimerMark() for _ in range(3000)]
s = torch.cuda.Stream()
for m in fresh2: m.record(s.cuda_stream)
torch.cuda.synchronize()
print("marks recorded once beforehand:    enqueue %.2f us/launch, total %.2f" % run(20000, fresh2, 20))
