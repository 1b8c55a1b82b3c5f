#!/usr/bin/env python3
"""The run kernels' source in the lane emulator under AddressSanitizer + UBSan (CPU only: the GPU pool has no sanitizer runs): the
one-channel-per-wave kernels (both block orders) and the run-to-run rendezvous of the 16-wave kernels (both block orders) on
exactly-sized buffers, LDS included (the emulator's LDS is a heap block).  tools/asan_emu.sh builds the library and runs this."""
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
emu = E(); o = orc.load()
def run(S,T,layout,seam,run8,rv,seed=9):
    wl = W.make_batch(n_streams=S, n_frames=T, layout=layout, mix=True, intensity=True, seed=seed)
    Cn = wl["C"]
    ov = np.zeros((S,Cn,1024),np.float32)
    ref, spec = o.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    coeffs, meta = (wl["q"], wl["meta"]) if seam=="q" else (spec.astype(np.float32), None)
    # exactly-sized copies so that ASan sees any access past the end
    units = wl["units"].copy(); coeffs = coeffs.copy(); meta = None if meta is None else meta.copy()
    pool = np.zeros((S,Cn,2,1024),np.float32); par=np.zeros(S*Cn,np.uint8)
    got = emu.decode(units, coeffs, meta, wl["n_pcm"], pool, par, run8=run8, rv=rv)
    err = float(np.sqrt(np.mean((got.astype(np.float64)-ref)**2)))
    print(layout,S,T,seam,"run8",run8,"rv",rv,"rms",err); assert err < 1e-5
for r8 in (1,2):
    run(2,20,("cpe",),"q",r8,1); run(1,19,("cpe",),"f",r8,1); run(1,35,("sce",),"q",r8,1); run(1,9,("cpe","cpe","cpe","sce"),"q",r8,1)
for rv in (1,2):
    run(2,37,("cpe",),"q",0,rv); run(1,33,("cpe",),"f",0,rv); run(1,50,("sce",),"q",0,rv); run(1,20,("cpe","cpe","cpe","sce"),"q",0,rv)
print("sanitized emulator runs ok")
