#!/usr/bin/env python3
"""The run kernels' source in the lane emulator under AddressSanitizer + UBSan (CPU only: the GPU pool has no sanitizer runs): the
run-to-run rendezvous of the 16-wave kernels (both block orders) and the launch-to-launch rendezvous of aacg_decode_pipelined
(three workgroup orders) on exactly-sized buffers, LDS included (the emulator's LDS is a heap block).  tools/asan_emu.sh builds
the library and runs this."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'aac.js_amd/python')
import emu_lib, orc, aacgpu_workload as W
# load the sanitized emulator library in place of the normal one
class E(emu_lib.Emu):
    def __init__(self):
        import subprocess
        self.lib = L = C.CDLL('/tmp/libaacg_emu_asan.so')
        L.emu_last_error.restype = C.c_char_p
        L.emu_decode_cce.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.emu_decode_pipelined.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]
emu = E(); o = orc.load()
def run(S,T,layout,seam,rv,seed=9):
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=seed)
    Cn = wl["C"]
    ov = np.zeros((S,Cn,1024),np.float32)
    ref, spec = o.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    coeffs, meta = (wl["q"], wl["meta"]) if seam=="q" else (spec.astype(np.float32), None)
    # exactly-sized copies so that ASan sees any access past the end
    units = wl["units"].copy(); coeffs = coeffs.copy(); meta = None if meta is None else meta.copy()
    pool, par = emu_lib.new_pool(S, Cn)
    got = emu.decode(units, coeffs, meta, wl["n_pcm"], pool, par, rv=rv)
    err = float(np.sqrt(np.mean((got.astype(np.float64)-ref)**2)))
    print(layout,S,T,seam,"rv",rv,"rms",err); assert err < 1e-5
def run_pipelined(S,T,layout,seam,order,n=3,seed=9):
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=seed)
    Cn = wl["C"]
    ov = np.zeros((S,Cn,1024),np.float32)
    refs, coeffs = [], []
    for j in range(n):
        q = np.roll(wl["q"], 3*j, axis=0).copy()
        ref, spec = o.decode_batch(wl["units"], q, wl["meta"], wl["n_pcm"], ov, want_spec=True)
        refs.append(ref); coeffs.append(q if seam=="q" else spec.astype(np.float32))
    pool, par = emu_lib.new_pool(S, Cn)
    cells = np.zeros((S,Cn,emu_lib.OV_BUFFERS,4),np.uint64); heads = np.zeros((S,Cn,emu_lib.OV_BUFFERS,1024),np.float32)
    got,_ = emu.decode_pipelined(wl["units"].copy(), coeffs, [wl["meta"].copy()]*n if seam=="q" else None, wl["n_pcm"], pool, par, cells, heads, order=order)
    for j in range(n):
        err = float(np.sqrt(np.mean((got[j].astype(np.float64)-refs[j])**2)))
        print("pipelined",layout,S,T,seam,"order",order,"launch",j,"rms",err); assert err < 1e-5
for rv in (1,2):
    run(2,37,("cpe",),"q",rv); run(1,33,("cpe",),"f",rv); run(1,50,("sce",),"q",rv); run(1,20,("cpe","cpe","cpe","sce"),"q",rv)
for order in (0,1,5):
    run_pipelined(2,16,("cpe",),"q",order); run_pipelined(1,20,("cpe",),"f",order); run_pipelined(1,4,("cpe","cpe","cpe","sce"),"q",order); run_pipelined(2,7,("sce",),"q",order)
print("sanitized emulator runs ok")
