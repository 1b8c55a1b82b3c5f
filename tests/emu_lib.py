"""ctypes wrapper for tests/emu/libaacg_emu.so — the kernels' source run lane-by-lane on CPU threads
(test infrastructure; see tests/emu/devport_emu.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
RUN_DTYPE = np.dtype([("pred_unit", "<i4"), ("n_units", "<i4"), ("is_last", "<i4"), ("wave_nch", "<u4"),
                      ("ov0", "<i4", (2,)), ("rot", "<i4", (2,)), ("unit", "<i4", (16,)),
                      ("wave_unit", "<i4", (16,)), ("wave_coef", "<u4", (16,)), ("wave_meta", "<u4", (16,))])     # aacg_run, aacg_device.h


OV_BUFFERS = 16         # AACG_OV_BUFFERS: rotating overlap buffers per channel (aacg_device.h)


def new_pool(streams, channels):
    """An engine's overlap pool, zeroed: [stream][channel][OV_BUFFERS][1024], and the live-buffer index per (stream, channel)."""
    return np.zeros((streams, channels, OV_BUFFERS, 1024), np.float32), np.zeros(streams * channels, np.uint8)


class Emu:
    def __init__(self, target="libaacg_emu.so"):
        import fcntl
        with open(os.path.join(HERE, "emu", ".build.lock"), "w") as lock:      # one make at a time (pytest-xdist workers)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.run(["make", "-C", os.path.join(HERE, "emu"), target], check=True, stdout=subprocess.DEVNULL)
        self.lib = L = C.CDLL(os.path.join(HERE, "emu", target))
        L.emu_last_error.restype = C.c_char_p
        L.emu_decode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.emu_decode_tns.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.emu_decode_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.emu_decode_cce.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.emu_spectral.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.emu_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.emu_plan_refresh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emu_decode_pipelined.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]
        L.emu_get_windows.argtypes = [C.c_int, C.c_void_p]
        L.emu_get_iq_sf.argtypes = [C.c_void_p, C.c_void_p]

    def error(self):
        return self.lib.emu_last_error().decode()

    def decode(self, units, coeffs, meta, n_pcm, pool, parity, sample_index=3, tns=None, pns=False, int16_out=False, cce=None, staged=False, unfused=False, rv=1):
        """staged: the optional stages (TNS, PNS) as a launch of their own even where the engine would run them inside the run kernel.
        unfused: independent coupling as the separate pass over the PCM (what plans with double-duty runs take) even where the
        engine applies it in the targets' epilogues."""
        units = np.ascontiguousarray(units)
        coeffs = np.ascontiguousarray(coeffs)
        kind = 1 if coeffs.dtype == np.int16 else 0
        meta = np.ascontiguousarray(meta, np.uint16) if meta is not None else None
        pcm = np.full(n_pcm, -32768, np.int16) if int16_out else np.full(n_pcm, np.nan, np.float32)
        tns = np.ascontiguousarray(tns) if tns is not None else None
        self.lib.emu_set_output_kind(1 if int16_out else 0)
        self.lib.emu_set_staged(1 if staged else 0)
        self.lib.emu_set_unfused(1 if unfused else 0)
        self.lib.emu_set_rv(rv)           # 1: chains longer than a run through the run-to-run rendezvous (the engine's route); 2: blocks in reverse; 0: recomputed frames
        cce = np.ascontiguousarray(cce) if cce is not None else None
        rc = self.lib.emu_decode_cce(kind, sample_index, pool.shape[0], pool.shape[1], units.ctypes.data, len(units),
                                     coeffs.ctypes.data, meta.ctypes.data if meta is not None else None,
                                     tns.ctypes.data if tns is not None else None, len(tns) if tns is not None else 0,
                                     1 if pns else 0, cce.ctypes.data if cce is not None else None, len(cce) if cce is not None else 0,
                                     pcm.ctypes.data, n_pcm, pool.ctypes.data, parity.ctypes.data)
        self.lib.emu_set_output_kind(0)
        self.lib.emu_set_staged(0)
        self.lib.emu_set_unfused(0)
        self.lib.emu_set_rv(1)
        if rc:
            raise RuntimeError("emu_decode rc=%d: %s" % (rc, self.error()))
        return pcm

    def spectral(self, units, q, meta, sample_index=3):
        units = np.ascontiguousarray(units)
        q = np.ascontiguousarray(q, np.int16)
        meta = np.ascontiguousarray(meta, np.uint16)
        spec = np.zeros(q.shape, np.float32)
        rc = self.lib.emu_spectral(sample_index, units.ctypes.data, len(units), q.ctypes.data, meta.ctypes.data, spec.ctypes.data)
        assert rc == 0
        return spec

    def plan_refresh(self, first, nxt, max_streams, max_channels, tns_spec=False, sample_index=3):
        """The host planner's refresh of a kept plan (aacg_plan_refresh_host): 0 or the AACG_ERR_* code."""
        first, nxt = np.ascontiguousarray(first), np.ascontiguousarray(nxt)
        assert len(first) == len(nxt)
        return self.lib.emu_plan_refresh(first.ctypes.data, nxt.ctypes.data, len(first), sample_index, max_streams, max_channels, 1 if tns_spec else 0)

    def decode_pipelined(self, units, coeffs_list, meta_list, n_pcm, pool, parity, cells, heads, order=0, epoch_in=0, sample_index=3, streams=0):
        """aacg_decode_pipelined for len(coeffs_list) consecutive launches of ONE plan, workgroup by workgroup in an order the
        engine's rules allow (order: 0 launch after launch, 1 the last launch of every round of streams first, >= 2 random interleaving
        with that seed; streams: how many the sequence takes in turn, 0 = the engine's choice for the plan).  cells: uint64 [S][C][OV_BUFFERS][4] (aacg_xl_cell), heads: like pool.  Returns (list of PCM arrays, epoch of the
        last launch: pass it as epoch_in to continue the sequence with the last launch 'still in flight')."""
        units = np.ascontiguousarray(units)
        n = len(coeffs_list)
        coeffs_list = [np.ascontiguousarray(c) for c in coeffs_list]
        kind = 1 if coeffs_list[0].dtype == np.int16 else 0
        meta_list = [np.ascontiguousarray(m, np.uint16) for m in meta_list] if meta_list is not None else None
        pcm = [np.full(n_pcm, np.nan, np.float32) for _ in range(n)]
        cp = (C.c_void_p * n)(*[c.ctypes.data for c in coeffs_list])
        mp = (C.c_void_p * n)(*[m.ctypes.data for m in meta_list]) if meta_list is not None else None
        pp = (C.c_void_p * n)(*[x.ctypes.data for x in pcm])
        last = C.c_uint64(0)
        rc = self.lib.emu_decode_pipelined(kind, sample_index, pool.shape[0], pool.shape[1], units.ctypes.data, len(units), n, cp, mp, pp, n_pcm,
                                           pool.ctypes.data, parity.ctypes.data, cells.ctypes.data, heads.ctypes.data, order, epoch_in, C.byref(last), streams)
        if rc:
            raise RuntimeError("emu_decode_pipelined rc=%d: %s" % (rc, self.error()))
        return pcm, last.value

    def plan(self, units, max_streams, max_channels, parity=None, sample_index=3):
        units = np.ascontiguousarray(units)
        runs = np.zeros(len(units) + 8, RUN_DTYPE)
        info = np.zeros(4, np.int32)
        n = self.lib.emu_plan(units.ctypes.data, len(units), sample_index, max_streams, max_channels,
                              parity.ctypes.data if parity is not None else None, runs.ctypes.data, len(runs), info.ctypes.data)
        return n, runs[:max(n, 0)], info

    def windows(self, sample_index=3):
        a = np.zeros(2304, np.float32)
        assert self.lib.emu_get_windows(sample_index, a.ctypes.data) == 0
        return a[:1024], a[1024:2048], a[2048:2176], a[2176:]

    def iq_sf(self):
        iq = np.zeros(8192, np.float32); sf = np.zeros(428, np.float32)
        assert self.lib.emu_get_iq_sf(iq.ctypes.data, sf.ctypes.data) == 0
        return iq, sf


def pool_current(pool, parity):
    """[S][C][1024] copy of the live overlap buffers at the reference's scale (the pool is PCM-scaled, like the engine's:
    aacg_get_overlap multiplies by 32768 too)."""
    S, Cn = pool.shape[:2]
    out = np.empty((S, Cn, 1024), np.float32)
    for s in range(S):
        for c in range(Cn):
            out[s, c] = pool[s, c, parity[s * Cn + c]] * np.float32(32768.0)
    return out


def emu_parse(emu, sample_index, entries, counts, data, frames, max_units, max_channels, options, want_tns):
    """The device front end's kernel source (aacg_parse.h) run on the CPU: same outputs as aacgpu.Parser.parse_batch."""
    import aacgpu
    data = np.ascontiguousarray(data, np.uint8)
    frames = np.ascontiguousarray(frames)
    entries = np.ascontiguousarray(entries)
    counts = np.ascontiguousarray(counts, np.uint32)
    out = aacgpu.alloc_parse_outputs(len(frames), max_units, max_channels, want_tns)
    f = emu.lib.emu_parse
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(sample_index, entries.ctypes.data, counts.ctypes.data, data.ctypes.data, data.size, frames.ctypes.data, len(frames),
           max_units, max_channels, options, out["units"].ctypes.data, out["q"].ctypes.data, out["meta"].ctypes.data,
           out["tns"].ctypes.data if want_tns else None, out["results"].ctypes.data)
    if rc:
        raise RuntimeError("emu_parse rc=%d: %s" % (rc, emu.error()))
    return out
