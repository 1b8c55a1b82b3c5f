"""ctypes wrapper around oracle/liboracle.so — the CPU checker (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

UNIT_DTYPE = np.dtype([
    ("stream", "<u4"), ("pcm_offset", "<u4"), ("channel", "<u2"), ("n_out_ch", "<u2"),
    ("n_ch", "u1"), ("flags", "u1"), ("reserved0", "<u2"), ("coef_offset", "<u4"), ("meta_offset", "<u4"),
    ("ch", [("window_sequence", "u1"), ("window_shape", "u1"), ("window_shape_prev", "u1"), ("max_sfb", "u1"),
            ("group_count", "u1"), ("flags", "u1"), ("reserved", "u1", (2,)), ("group_len", "u1", (8,))], (2,)),
    ("tns_offset", "<u4"), ("reserved1", "<u4"),
])
assert UNIT_DTYPE.itemsize == 64
TNS_DTYPE = np.dtype([
    ("n_filt", "u1", (8,)),
    ("filt", [("length", "u1"), ("order", "u1"), ("direction", "u1"), ("reserved", "u1"), ("coef", "<f4", (12,))], (8,)),
])
assert TNS_DTYPE.itemsize == 424
CHAN_INFO_DTYPE = UNIT_DTYPE["ch"].base
CCE_DTYPE = np.dtype([("coupling_point", "u1"), ("n_targets", "u1"), ("reserved", "u1", (2,)),
                      ("target", [("channel", "u1"), ("gain_list", "u1")], (16,)), ("gain", "<f4", (16, 120))])
assert CCE_DTYPE.itemsize == 7716


def build(target="liboracle.so"):
    # one make at a time: pytest-xdist workers (and bench.py's ranks) all come here, and two makes writing one .so is a torn file
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", ORACLE_DIR, target], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE_DIR, target)


class Oracle:
    def __init__(self, path):
        self.lib = L = C.CDLL(path)
        L.orc_init.restype = None
        L.orc_get_table_f32.restype = C.c_size_t
        L.orc_get_table_f32.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.orc_get_table_f64.restype = C.c_size_t
        L.orc_get_table_f64.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.orc_get_swb_offsets.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_fft_inverse.restype = None
        L.orc_fft_inverse.argtypes = [C.c_int, C.c_void_p]
        L.orc_imdct.restype = None
        L.orc_imdct.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_filterbank.restype = None
        L.orc_filterbank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pns_sequence.restype = None
        L.orc_pns_sequence.argtypes = [C.c_void_p, C.c_int]
        L.orc_decode_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode_batch_tns.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode_batch_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tns_spec.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_decode_batch_cce.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bench_threads.restype = C.c_longlong
        L.orc_bench_threads.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
        L.orc_init()

    def table_f32(self, which):
        n = self.lib.orc_get_table_f32(which, None, 0)
        a = np.empty(n, np.float32)
        self.lib.orc_get_table_f32(which, a.ctypes.data, n)
        return a

    def table_f64(self, which):
        n = self.lib.orc_get_table_f64(which, None, 0)
        a = np.empty(n, np.float64)
        self.lib.orc_get_table_f64(which, a.ctypes.data, n)
        return a

    def swb_offsets(self, sample_index, is_long):
        a = np.zeros(64, np.uint16)
        n = self.lib.orc_get_swb_offsets(sample_index, int(is_long), a.ctypes.data)
        return a[:n + 1].copy()

    def exact_fft_roots(self):
        """Context manager: inside it the oracle's FFT uses correctly rounded roots instead of the reference's float32
        recurrence (orc_set_fft_roots) — for attributing the GPU engine's distance from the reference."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.lib.orc_set_fft_roots(1)
            try:
                yield self
            finally:
                self.lib.orc_set_fft_roots(0)
        return cm()

    def fft_inverse(self, x):
        buf = np.ascontiguousarray(x, np.float32).copy()
        self.lib.orc_fft_inverse(buf.shape[0], buf.ctypes.data)
        return buf

    def imdct(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty(2 * x.shape[0], np.float32)
        self.lib.orc_imdct(2 * x.shape[0], x.ctypes.data, y.ctypes.data)
        return y

    def filterbank(self, seq, shape, shape_prev, x, overlap):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(1024, np.float32)
        self.lib.orc_filterbank(seq, shape, shape_prev, x.ctypes.data, out.ctypes.data, overlap.ctypes.data)
        return out

    def pns_sequence(self, n):
        a = np.empty(n, np.int32)
        self.lib.orc_pns_sequence(a.ctypes.data, n)
        return a

    def tns_spec(self, info, tns, data, sample_index=3):
        """AACG_TNS_SPEC on one channel: info a CHAN_INFO_DTYPE scalar, tns a TNS_DTYPE scalar; returns filtered copy."""
        info = np.array(info, CHAN_INFO_DTYPE).reshape(1)
        tns = np.array(tns, TNS_DTYPE).reshape(1)
        out = np.array(data, np.float32).copy()
        assert out.size == 1024
        rc = self.lib.orc_tns_spec(sample_index, info.ctypes.data, tns.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_tns_spec failed: %d" % rc)
        return out

    def bench_threads(self, n_threads, seconds, units, coeffs, meta, n_pcm_floats, max_streams, max_channels, sample_index=3):
        """CPU-baseline timing loop (oracle/orc_bench.c): returns (whole batches decoded by all threads, longest thread's seconds)."""
        units, coeffs = np.ascontiguousarray(units), np.ascontiguousarray(coeffs)
        kind = 1 if coeffs.dtype == np.int16 else 0
        if meta is not None:
            meta = np.ascontiguousarray(meta, np.uint16)
        dt = C.c_double()
        n = self.lib.orc_bench_threads(n_threads, seconds, sample_index, kind, max_streams, max_channels, units.ctypes.data, len(units),
                                       coeffs.ctypes.data, coeffs.nbytes, meta.ctypes.data if meta is not None else None,
                                       meta.size // 120 if meta is not None else 0, n_pcm_floats, C.byref(dt))
        if n < 0:
            raise RuntimeError("orc_bench_threads failed: %d" % n)
        return int(n), float(dt.value)

    def decode_batch(self, units, coeffs, meta, n_pcm_floats, overlaps, sample_index=3, want_spec=False, tns=None, pns=False, cce=None):
        """overlaps: float32 [max_streams, max_channels, 1024], updated in place.  tns: TNS_DTYPE array -> AACG_TNS_SPEC."""
        units = np.ascontiguousarray(units)
        assert units.dtype == UNIT_DTYPE
        coeffs = np.ascontiguousarray(coeffs)
        kind = 1 if coeffs.dtype == np.int16 else 0
        assert kind == 1 or coeffs.dtype == np.float32
        if meta is not None:
            meta = np.ascontiguousarray(meta, np.uint16)
        pcm = np.full(n_pcm_floats, np.nan, np.float32)
        spec = np.zeros(coeffs.size, np.float32) if want_spec else None
        assert overlaps.dtype == np.float32 and overlaps.flags.c_contiguous
        if tns is not None:
            tns = np.ascontiguousarray(tns)
            assert tns.dtype == TNS_DTYPE
        if cce is not None:
            cce = np.ascontiguousarray(cce)
            assert cce.dtype == CCE_DTYPE
        rc = self.lib.orc_decode_batch_cce(sample_index, kind, overlaps.shape[0], overlaps.shape[1],
                                           units.ctypes.data, len(units), coeffs.ctypes.data,
                                           meta.ctypes.data if meta is not None else None,
                                           tns.ctypes.data if tns is not None else None, 1 if tns is not None else 0,
                                           1 if pns else 0,
                                           cce.ctypes.data if cce is not None else None, len(cce) if cce is not None else 0,
                                           pcm.ctypes.data, overlaps.ctypes.data,
                                           spec.ctypes.data if want_spec else None)
        if rc != 0:
            raise RuntimeError("orc_decode_batch failed: %d" % rc)
        return (pcm, spec.reshape(-1, 1024)) if want_spec else pcm


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(build())
    return _cached
