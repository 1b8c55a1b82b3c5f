import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, "aac.js_amd/python")
import aacgpu
L = aacgpu.load_library()
class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("abi_version","device_ordinal","sample_index","max_streams","channels","max_frames","output_kind","parse_options")]
entries, counts = aacgpu.standard_codebooks()
one = np.fromfile("tests/golden/streams/stereo48.aac", np.uint8)
# frame table of one stream
offs, at = [], 0
while at + 7 <= len(one):
    ln = ((int(one[at+3]) & 3) << 11) | (int(one[at+4]) << 3) | (int(one[at+5]) >> 5)
    offs.append((at, ln)); at += ln
S, F = 256, 16
per = sum(l for _, l in offs[:F])
bytes_ = np.tile(one[:per], S)
frames = np.zeros((S*F, 2), np.uint32)
for s in range(S):
    a = s * per
    for f in range(F):
        frames[s*F+f] = (a, offs[f][1]); a += offs[f][1]
slots = np.arange(S, dtype=np.uint32)
cfg = Cfg(L.aacg_abi_version(), 0, 3, S, 2, F, 0, 2)
h = C.c_void_p()
L.aacg_pipeline_create.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
L.aacg_pipeline_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
L.aacg_pipeline_last_error.restype = C.c_char_p; L.aacg_pipeline_last_error.argtypes = [C.c_void_p]
t0 = time.perf_counter(); rc = L.aacg_pipeline_create(C.byref(cfg), entries.ctypes.data, counts.ctypes.data, C.byref(h)); print("create rc", rc, "%.1f ms" % ((time.perf_counter()-t0)*1e3))
pcm = np.zeros(S*F*2048, np.float32); res = np.zeros(S*F*8, np.uint8); refused = C.c_uint32(0)
def run(dst, n=10):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        rc = L.aacg_pipeline_decode(h, bytes_.ctypes.data, bytes_.size, frames.ctypes.data, slots.ctypes.data, S, F, dst, res.ctypes.data, C.byref(refused))
        ts.append((time.perf_counter()-t0)*1e3)
        assert rc == 0, L.aacg_pipeline_last_error(h)
    return ts
print("pageable out:", ["%.2f" % t for t in run(pcm.ctypes.data)])
t0 = time.perf_counter(); p = L.aacg_host_alloc(pcm.nbytes); print("host_alloc 33.5 MB: %.2f ms" % ((time.perf_counter()-t0)*1e3))
print("pinned out:  ", ["%.2f" % t for t in run(p)])
print("refused", refused.value, "pcm rms", float(np.sqrt(np.mean(pcm.astype(np.float64)**2))))
