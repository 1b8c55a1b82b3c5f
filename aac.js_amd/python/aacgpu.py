"""ctypes binding of the C ABI in include/aacgpu.h (aac.js_amd/csrc/libaacgpu.so).

Plumbing only: tests and bench.py drive the HIP path through exactly the entry points a
JavaScript host binds over N-API.  There is no CPU fallback here: if the
library is missing or no GPU is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# AACGPU_LIB: A/B another build of the same ABI (tools/ab.sh); default is the in-tree build
LIB_PATH = os.environ.get("AACGPU_LIB") or os.path.join(ROOT, "aac.js_amd", "csrc", "libaacgpu.so")

INPUT_SPEC_F32, INPUT_QUANT_I16 = 0, 1
OUTPUT_F32, OUTPUT_I16 = 0, 1
ERR_NAMES = {0: "OK", -1: "INVALID_ARG", -2: "NO_DEVICE", -3: "OUT_OF_MEMORY", -4: "CAPACITY",
             -5: "UNSUPPORTED", -6: "LAYOUT_CHANGE", -7: "STALE_PLAN", -8: "TIMEOUT"}
ERR_TIMEOUT = -8

# every symbol include/aacgpu.h declares
ABI_SYMBOLS = ["aacg_create", "aacg_destroy", "aacg_last_error", "aacg_abi_version", "aacg_reset_stream",
               "aacg_get_overlap", "aacg_set_overlap", "aacg_decode_batch", "aacg_submit", "aacg_wait",
               "aacg_decode_batch_tns", "aacg_submit_tns", "aacg_plan_create_tns",
               "aacg_host_alloc", "aacg_host_free", "aacg_plan_create", "aacg_plan_destroy",
               "aacg_decode_device", "aacg_spectral_device", "aacg_synchronize", "aacg_get_table", "aacg_kernel_name",
               "aacg_parser_create", "aacg_parser_destroy", "aacg_parser_last_error", "aacg_parse_status_string",
               "aacg_parse_batch", "aacg_parse_device", "aacg_parse_kernel_name", "aacg_plan_refresh_from_parse", "aacg_standard_codebooks",
               "aacg_decode_batch_ex", "aacg_submit_ex", "aacg_plan_create_ex", "aacg_plan_kernels", "aacg_plan_kernels_ex", "aacg_plan_refresh_units",
               "aacg_decode_pipelined", "aacg_pipeline_fork", "aacg_pipeline_join",
               "aacg_pipeline_create", "aacg_pipeline_destroy", "aacg_pipeline_last_error", "aacg_pipeline_reset_stream", "aacg_pipeline_decode",
               "aacg_pipeline_submit", "aacg_pipeline_collect", "aacg_pipeline_set_wait_limit_ms", "aacg_pipeline_stream_layout",
               "aacg_set_wait_limit_ms", "aacg_parser_set_wait_limit_ms", "aacg_pipeline_info",
               "aacg_plan_set_unit_sets", "aacg_plan_refresh_from_parse_ex"]
# ... and include/aacgpu_tools.h (measurement and diagnostics: bench.py, tools/, tests)
TOOLS_SYMBOLS = ["aacg_calib_copy", "aacg_timer_create", "aacg_timer_record", "aacg_timer_elapsed_ms", "aacg_timer_destroy",
                 "aacg_pipeline_chained", "aacg_pipeline_concurrent", "aacg_decode_pipelined_timed", "aacg_debug_transform", "aacg_debug_set_route", "aacg_debug_route", "aacg_debug_run_kernel",
                 "aacg_debug_pipeline_order", "aacg_pipeline_streams_used", "aacg_debug_set_wait_mode", "aacg_debug_in_flight", "aacg_debug_stall"]
WAIT_SPIN, WAIT_YIELD, WAIT_SLEEP, WAIT_BLOCK = 0, 1, 2, 3
# aacg_debug_set_route / aacg_debug_route flags
DEBUG_ROUTE_UNFUSED_COUPLING, DEBUG_ROUTE_RECOMPUTE = 1, 8
ROUTE_PLAN_TNS, ROUTE_PLAN_PNS, ROUTE_PLAN_LONG_CHAINS, ROUTE_PLAN_FULL_LATER_RUNS = 1, 2, 4, 8
ROUTE_PLAN_WIDE_FRAMES, ROUTE_PLAN_CCE_INDEPENDENT, ROUTE_PLAN_CCE_DEPENDENT, ROUTE_PLAN_NO_RUNS = 0x10, 0x20, 0x40, 0x80
# switches of a run kernel (aacg_routes.h), as aacg_debug_run_kernel returns them
RK_QUANT, RK_I16, RK_DD, RK_EX, RK_CPL, RK_RV, RK_NT = 1, 2, 4, 8, 16, 32, 64

UNIT_DTYPE = np.dtype([
    ("stream", "<u4"), ("pcm_offset", "<u4"), ("channel", "<u2"), ("n_out_ch", "<u2"),
    ("n_ch", "u1"), ("flags", "u1"), ("reserved0", "<u2"), ("coef_offset", "<u4"), ("meta_offset", "<u4"),
    ("ch", [("window_sequence", "u1"), ("window_shape", "u1"), ("window_shape_prev", "u1"), ("max_sfb", "u1"),
            ("group_count", "u1"), ("flags", "u1"), ("reserved", "u1", (2,)), ("group_len", "u1", (8,))], (2,)),
    ("tns_offset", "<u4"), ("reserved1", "<u4"),
])
assert UNIT_DTYPE.itemsize == 64

# aacg_tns_info: one per channel with CHAN_TNS_PRESENT (TNS_SPEC engines)
TNS_DTYPE = np.dtype([
    ("n_filt", "u1", (8,)),
    ("filt", [("length", "u1"), ("order", "u1"), ("direction", "u1"), ("reserved", "u1"), ("coef", "<f4", (12,))], (8,)),
])
assert TNS_DTYPE.itemsize == 424
# aacg_cce_info: one per coupling channel element (CCE_SPEC engines; units flagged UNIT_CCE, reserved1 = its index)
CCE_DTYPE = np.dtype([("coupling_point", "u1"), ("n_targets", "u1"), ("reserved", "u1", (2,)),
                      ("target", [("channel", "u1"), ("gain_list", "u1")], (16,)), ("gain", "<f4", (16, 120))])
assert CCE_DTYPE.itemsize == 7716
CCE_REFERENCE, CCE_SPEC = 0, 1
CCE_BEFORE_TNS, CCE_AFTER_TNS, CCE_AFTER_IMDCT = 0, 1, 2
TNS_REFERENCE, TNS_SPEC = 0, 1
PNS_REFERENCE, PNS_SPEC = 0, 1
UNIT_COMMON_WINDOW, UNIT_MASK_PRESENT, UNIT_HAS_PNS, UNIT_CCE = 1, 2, 4, 8
CHAN_TNS_PRESENT = 0x01


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_ordinal", C.c_int32), ("sample_index", C.c_int32),
                ("max_streams", C.c_int32), ("max_channels", C.c_int32), ("max_batch_units", C.c_int32),
                ("input_kind", C.c_int32), ("tns_mode", C.c_int32), ("pns_mode", C.c_int32), ("output_kind", C.c_int32), ("cce_mode", C.c_int32)]


class PipelineConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_ordinal", C.c_int32), ("sample_index", C.c_int32), ("max_streams", C.c_int32),
                ("channels", C.c_int32), ("max_frames", C.c_int32), ("output_kind", C.c_int32), ("parse_options", C.c_int32),
                ("lanes", C.c_int32), ("reserved", C.c_int32 * 3)]


class Batch(C.Structure):
    _fields_ = [("units", C.c_void_p), ("n_units", C.c_uint32), ("coeffs", C.c_void_p), ("n_coef_blocks", C.c_uint32),
                ("meta", C.c_void_p), ("n_meta", C.c_uint32), ("tns", C.c_void_p), ("n_tns", C.c_uint32),
                ("cce", C.c_void_p), ("n_cce", C.c_uint32), ("pcm_out", C.c_void_p), ("n_pcm_floats", C.c_size_t)]


class AacgError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "aacgpu: %s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7.  One process must use ONE HIP runtime, or device
    pointers from torch are foreign to this library: if torch is installed (bench/tests use it for device
    memory and streams) map its runtime first, so that libaacgpu.so's NEEDED libamdhip64.so.7 resolves to
    the same copy whether torch is imported before or after.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library.  Raises if it has not been built — never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    L.aacg_abi_version.restype = C.c_int
    L.aacg_kernel_name.restype = C.c_char_p
    L.aacg_plan_kernels.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.aacg_plan_kernels_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.aacg_decode_pipelined.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_decode_pipelined_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_pipeline_fork.argtypes = [C.c_void_p, C.c_void_p]
    L.aacg_pipeline_join.argtypes = [C.c_void_p, C.c_void_p]
    L.aacg_pipeline_chained.argtypes = [C.c_void_p]
    L.aacg_pipeline_chained.restype = C.c_uint64
    L.aacg_pipeline_concurrent.argtypes = [C.c_void_p]
    L.aacg_pipeline_streams_used.argtypes = [C.c_void_p]
    L.aacg_debug_route.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.aacg_debug_run_kernel.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.aacg_plan_refresh_units.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.aacg_calib_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.aacg_timer_create.argtypes = [C.POINTER(C.c_void_p)]
    L.aacg_timer_record.argtypes = [C.c_void_p, C.c_void_p]
    L.aacg_timer_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.aacg_timer_destroy.argtypes = [C.c_void_p]
    L.aacg_timer_destroy.restype = None
    L.aacg_debug_set_route.argtypes = [C.c_void_p, C.c_int]
    L.aacg_last_error.restype = C.c_char_p
    L.aacg_last_error.argtypes = [C.c_void_p]
    L.aacg_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.aacg_destroy.argtypes = [C.c_void_p]
    L.aacg_destroy.restype = None
    L.aacg_reset_stream.argtypes = [C.c_void_p, C.c_uint32]
    L.aacg_get_overlap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.aacg_set_overlap.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.aacg_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.aacg_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.aacg_decode_batch_tns.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.aacg_submit_tns.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                  C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_uint64)]
    L.aacg_plan_create_tns.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.aacg_wait.argtypes = [C.c_void_p, C.c_uint64]
    L.aacg_host_alloc.restype = C.c_void_p
    L.aacg_host_alloc.argtypes = [C.c_size_t]
    L.aacg_host_free.restype = None
    L.aacg_host_free.argtypes = [C.c_void_p]
    L.aacg_plan_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.aacg_plan_destroy.argtypes = [C.c_void_p]
    L.aacg_plan_destroy.restype = None
    L.aacg_decode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_spectral_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_synchronize.argtypes = [C.c_void_p, C.c_void_p]
    L.aacg_get_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.aacg_parser_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.aacg_parser_destroy.argtypes = [C.c_void_p]
    L.aacg_parser_destroy.restype = None
    L.aacg_parser_last_error.argtypes = [C.c_void_p]
    L.aacg_parser_last_error.restype = C.c_char_p
    L.aacg_parse_status_string.argtypes = [C.c_int]
    L.aacg_parse_status_string.restype = C.c_char_p
    L.aacg_parse_kernel_name.restype = C.c_char_p
    L.aacg_parse_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_plan_refresh_from_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.aacg_parse_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_standard_codebooks.argtypes = [C.c_void_p, C.c_void_p]
    L.aacg_decode_batch_ex.argtypes = [C.c_void_p, C.POINTER(Batch)]
    L.aacg_submit_ex.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(C.c_uint64)]
    L.aacg_plan_create_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.aacg_debug_transform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.aacg_standard_codebooks.restype = C.c_uint32
    L.aacg_set_wait_limit_ms.argtypes = [C.c_void_p, C.c_uint32]
    L.aacg_parser_set_wait_limit_ms.argtypes = [C.c_void_p, C.c_uint32]
    L.aacg_pipeline_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.aacg_debug_set_wait_mode.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.aacg_debug_in_flight.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.aacg_plan_set_unit_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.aacg_plan_refresh_from_parse_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.aacg_pipeline_create.argtypes = [C.POINTER(PipelineConfig), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.aacg_pipeline_destroy.argtypes = [C.c_void_p]
    L.aacg_pipeline_destroy.restype = None
    L.aacg_pipeline_last_error.argtypes = [C.c_void_p]
    L.aacg_pipeline_last_error.restype = C.c_char_p
    L.aacg_pipeline_reset_stream.argtypes = [C.c_void_p, C.c_uint32]
    L.aacg_pipeline_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.aacg_pipeline_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.aacg_pipeline_collect.argtypes = [C.c_void_p, C.c_uint64]
    L.aacg_pipeline_set_wait_limit_ms.argtypes = [C.c_void_p, C.c_uint32]
    L.aacg_pipeline_stream_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    _lib = L
    return L


# ---- device front end (aacg_parser_*, include/aacgpu.h) -------------------------------------------------
CODE_ENTRY_DTYPE = np.dtype([("code", "<u4"), ("len", "u1"), ("v", "i1", (4,)), ("reserved", "u1", (3,))])
PARSE_FRAME_DTYPE = np.dtype([("byte_offset", "<u4"), ("byte_length", "<u4")])
PARSE_RESULT_DTYPE = np.dtype([("status", "u1"), ("n_units", "u1"), ("n_channels", "u1"), ("flags", "u1"), ("bits_used", "<u4")])
META_DTYPE = np.dtype(("<u2", (120,)))
PARSE_APPLY_PULSES, PARSE_REFERENCE_QUIRKS, PARSE_SKIP_ZERO_FILL = 1, 2, 4


def debug_transform(x, is_short=False, identity_rotation=False, sample_index=3, device=0, vm=False):
    """aacg_debug_transform: the kernels' IMDCT stage on one spectrum (1024 floats), windows forced to 1; returns 2048 floats.
    vm: the int16 seam's variant of the stage (mirror-lane exchanges as DPP, long columns dealt out by long_col)."""
    L = load_library()
    x = np.ascontiguousarray(x, np.float32)
    assert x.size == 1024
    out = np.zeros(2048, np.float32)
    rc = L.aacg_debug_transform(device, sample_index, int(bool(is_short)) | (2 if vm else 0), int(identity_rotation), x.ctypes.data, out.ctypes.data)
    if rc:
        raise AacgError(rc, "aacg_debug_transform failed")
    return out



class TimerMark:
    """aacg_timer_*: a HIP event that only measures time (hipEventDisableSystemFence), recorded on a given stream.
    `a.elapsed_ms(b)` after the stream has been synchronised."""
    def __init__(self):
        self.h = C.c_void_p()
        rc = load_library().aacg_timer_create(C.byref(self.h))
        if rc:
            raise AacgError(rc, "aacg_timer_create failed")

    def record(self, stream):
        rc = load_library().aacg_timer_record(self.h, stream)
        if rc:
            raise AacgError(rc, "aacg_timer_record failed")

    def elapsed_ms(self, later):
        ms = C.c_float()
        rc = load_library().aacg_timer_elapsed_ms(self.h, later.h, C.byref(ms))
        if rc:
            raise AacgError(rc, "aacg_timer_elapsed_ms failed (stream not synchronised?)")
        return float(ms.value)

    def __del__(self):
        try:
            if self.h:
                load_library().aacg_timer_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

def debug_route(input_kind, output_kind, plan_flags, pipelined=False, debug_flags=0):
    """aacg_debug_route: the launches (kernel names, ' + ' between them) the engine's ONE route decision makes for a planned batch
    with these flags (ROUTE_PLAN_*) on an engine of these kinds.  No device needed."""
    buf = C.create_string_buffer(512)
    rc = load_library().aacg_debug_route(input_kind, output_kind, debug_flags, plan_flags, 1 if pipelined else 0, buf, 512)
    if rc:
        raise AacgError(rc, "aacg_debug_route: no registered kernel for this route")
    return buf.value.decode()


def run_kernels():
    """The registered run kernels as {symbol: switches} (aacg_debug_run_kernel)."""
    out, i = {}, 0
    buf = C.create_string_buffer(128)
    while True:
        key = load_library().aacg_debug_run_kernel(i, buf, 128)
        if key < 0:
            return out
        out[buf.value.decode()] = key
        i += 1


PIPE_STREAMS = 3        # AACG_PIPE_STREAMS: launches of a pipelined sequence that may be in flight side by side (aacg_device.h)


def pipeline_order(n, streams=PIPE_STREAMS):
    """How launch n of a pipelined sequence on `streams` streams is ordered (aacg_pipeline_order, aacg_routes.cpp): (stream, the round whose completion
    events the host waits for before enqueuing it or -1, whether its own completion gets an event, the launch up to which
    everything is known complete when it is enqueued or -1, number of rotating overlap buffers the rule has to cover)."""
    lib = load_library()
    lib.aacg_debug_pipeline_order.argtypes = [C.c_ulonglong, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    st, w, m, u = C.c_int(), C.c_longlong(), C.c_int(), C.c_longlong()
    k = lib.aacg_debug_pipeline_order(n, streams, C.byref(st), C.byref(w), C.byref(m), C.byref(u))
    return st.value, w.value, bool(m.value), u.value, k


def calib_copy(d_dst, d_src, n_bytes, stream=0):
    """aacg_calib_copy: float4 device-to-device copy with the run kernels' launch shape, enqueued on `stream`."""
    rc = load_library().aacg_calib_copy(d_dst, d_src, n_bytes, stream)
    if rc:
        raise AacgError(rc, "aacg_calib_copy failed")


def standard_codebooks():
    """The 12 codebooks of ISO/IEC 14496-3 as (entries, counts), from the library (aacg_standard_codebooks)."""
    L = load_library()
    entries, counts = np.zeros(L.aacg_standard_codebooks(None, None), CODE_ENTRY_DTYPE), np.zeros(12, np.uint32)
    L.aacg_standard_codebooks(entries.ctypes.data, counts.ctypes.data)
    return entries, counts


def alloc_parse_outputs(n_frames, max_units, max_channels, want_tns):
    blocks = n_frames * max_channels
    return {"units": np.zeros(n_frames * max_units, UNIT_DTYPE), "q": np.zeros((blocks, 1024), np.int16),
            "meta": np.zeros((blocks, 120), np.uint16), "tns": np.zeros(blocks, TNS_DTYPE) if want_tns else None,
            "results": np.zeros(n_frames, PARSE_RESULT_DTYPE)}


class Parser:
    """aacg_parser: one GPU lane parses one frame.  entries / counts: the 12 codebooks as CODE_ENTRY_DTYPE records."""

    def __init__(self, entries=None, counts=None, sample_index=3, device=0):
        self.lib = load_library()
        if entries is None:
            entries, counts = standard_codebooks()
        entries = np.ascontiguousarray(entries)
        counts = np.ascontiguousarray(counts, np.uint32)
        assert entries.dtype == CODE_ENTRY_DTYPE and counts.size == 12
        h = C.c_void_p()
        rc = self.lib.aacg_parser_create(device, sample_index, entries.ctypes.data, counts.ctypes.data, C.byref(h))
        self.handle = h
        if rc != 0:
            msg = self.lib.aacg_parser_last_error(h).decode() if h else "aacg_parser_create failed"
            self.close()
            raise AacgError(rc, msg)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.aacg_parser_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AacgError(rc, self.lib.aacg_parser_last_error(self.handle).decode())

    def parse_batch(self, data, frames, max_units, max_channels, options=PARSE_REFERENCE_QUIRKS, want_tns=False):
        data = np.ascontiguousarray(data, np.uint8)
        frames = np.ascontiguousarray(frames)
        assert frames.dtype == PARSE_FRAME_DTYPE
        out = alloc_parse_outputs(len(frames), max_units, max_channels, want_tns)
        self._check(self.lib.aacg_parse_batch(self.handle, data.ctypes.data, data.size, frames.ctypes.data, len(frames), max_units, max_channels,
                                              options, out["units"].ctypes.data, out["q"].ctypes.data, out["meta"].ctypes.data,
                                              out["tns"].ctypes.data if want_tns else None, out["results"].ctypes.data))
        return out

    def parse_device(self, d_bytes, d_frames, n_frames, max_units, max_channels, options, d_units, d_q, d_meta, d_tns, d_results, stream=0):
        """Device pointers (ints); asynchronous on `stream`."""
        self._check(self.lib.aacg_parse_device(self.handle, d_bytes, d_frames, n_frames, max_units, max_channels, options,
                                               d_units, d_q, d_meta, d_tns, d_results, stream))

    def status_string(self, status):
        return self.lib.aacg_parse_status_string(int(status)).decode()


REFRESH_MAP_DTYPE = np.dtype([("parsed_index", "<u4"), ("frame_units", "<u4")])
PARSE_LAYOUT = 16          # AACG_PARSE_LAYOUT: a frame whose elements are not its stream's


class Pipeline:
    """aacg_pipeline: bytes in, PCM out — device front end and transform behind one call per batch, `lanes` batches in flight
    (include/aacgpu.h).  channels = the streams' chanConfig (1..8)."""

    def __init__(self, channels=2, max_streams=1, max_frames=16, sample_index=3, device=0, output_kind=OUTPUT_F32,
                 parse_options=PARSE_REFERENCE_QUIRKS, lanes=0, entries=None, counts=None):
        self.lib = load_library()
        if entries is None:
            entries, counts = standard_codebooks()
        entries, counts = np.ascontiguousarray(entries), np.ascontiguousarray(counts, np.uint32)
        cfg = PipelineConfig(self.lib.aacg_abi_version(), device, sample_index, max_streams, channels, max_frames, output_kind, parse_options, lanes)
        h = C.c_void_p()
        rc = self.lib.aacg_pipeline_create(C.byref(cfg), entries.ctypes.data, counts.ctypes.data, C.byref(h))
        if rc != 0:
            raise AacgError(rc, "aacg_pipeline_create failed (no GPU?)")
        self.handle, self.channels, self.i16 = h, channels, output_kind == OUTPUT_I16
        self._keep = {}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.aacg_pipeline_destroy(self.handle)
            self.handle = None
            for p in getattr(self, "_pinned", []):
                self.lib.aacg_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise AacgError(rc, self.lib.aacg_pipeline_last_error(self.handle).decode())

    def pinned(self, shape, dtype):
        """numpy array over page-locked host memory (aacg_host_alloc): the PCM comes straight down from the device into it."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.aacg_host_alloc(n)
        if not p:
            raise AacgError(-3, "aacg_host_alloc failed")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.frombuffer((C.c_char * n).from_address(p), dtype=dtype).reshape(shape)

    def set_wait_limit_ms(self, ms):
        self._check(self.lib.aacg_pipeline_set_wait_limit_ms(self.handle, int(ms)))

    def reset_stream(self, slot):
        self._check(self.lib.aacg_pipeline_reset_stream(self.handle, slot))

    def stream_layout(self, slot):
        """(channels of every element of the slot's frames, how many of the elements are decoded); ([], 0) before it is learnt."""
        ch, kept = np.zeros(8, np.uint8), C.c_uint32()
        n = self.lib.aacg_pipeline_stream_layout(self.handle, slot, ch.ctypes.data, C.byref(kept))
        if n < 0:
            raise AacgError(n, "aacg_pipeline_stream_layout")
        return [int(c) for c in ch[:n]], int(kept.value)

    def _args(self, data, frames, slots, frames_per_stream, pcm):
        data = np.ascontiguousarray(data, np.uint8)
        frames = np.ascontiguousarray(frames)
        slots = np.ascontiguousarray(slots, np.uint32)
        assert frames.dtype == PARSE_FRAME_DTYPE and len(frames) == len(slots) * frames_per_stream
        n = len(frames)
        if pcm is None:
            pcm = np.zeros(n * 1024 * self.channels, np.int16 if self.i16 else np.float32)
        assert pcm.size >= n * 1024 * self.channels
        return data, frames, slots, pcm, np.zeros(n, PARSE_RESULT_DTYPE), C.c_uint32(0)

    def decode(self, data, frames, slots, frames_per_stream, pcm=None):
        """Synchronous: (pcm, results, n_refused).  frames[s * F + f] = frame f of stream s in `data`."""
        data, frames, slots, pcm, res, refused = self._args(data, frames, slots, frames_per_stream, pcm)
        self._check(self.lib.aacg_pipeline_decode(self.handle, data.ctypes.data, data.size, frames.ctypes.data, slots.ctypes.data, len(slots),
                                                  frames_per_stream, pcm.ctypes.data, res.ctypes.data, C.byref(refused)))
        return pcm, res, int(refused.value)

    def submit(self, data, frames, slots, frames_per_stream, pcm=None):
        """Asynchronous: returns a ticket; collect(ticket) -> (pcm, results, n_refused)."""
        data, frames, slots, pcm, res, refused = self._args(data, frames, slots, frames_per_stream, pcm)
        t = C.c_uint64()
        self._check(self.lib.aacg_pipeline_submit(self.handle, data.ctypes.data, data.size, frames.ctypes.data, slots.ctypes.data, len(slots),
                                                  frames_per_stream, pcm.ctypes.data, res.ctypes.data, C.byref(refused), C.byref(t)))
        self._keep[t.value] = (pcm, res, refused)
        return t.value

    def collect(self, ticket):
        self._check(self.lib.aacg_pipeline_collect(self.handle, ticket))
        pcm, res, refused = self._keep.pop(ticket)
        return pcm, res, int(refused.value)


class Plan:
    def __init__(self, engine, handle, n_units):
        self.engine, self.handle, self.n_units = engine, handle, n_units

    def destroy(self):
        if self.handle:
            self.engine.lib.aacg_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Engine:
    """One engine per device (mirrors one FilterBank per decoder, for many streams at once)."""

    def __init__(self, input_kind=INPUT_QUANT_I16, max_streams=1, max_channels=2, device=0, sample_index=3,
                 max_batch_units=0, tns_mode=TNS_REFERENCE, pns_mode=PNS_REFERENCE, output_kind=OUTPUT_F32, cce_mode=CCE_REFERENCE):
        self.lib = load_library()
        cfg = Config(self.lib.aacg_abi_version(), device, sample_index, max_streams, max_channels, max_batch_units,
                     input_kind, tns_mode, pns_mode, output_kind, cce_mode)
        self.pcm_dtype = np.int16 if output_kind == OUTPUT_I16 else np.float32
        h = C.c_void_p()
        rc = self.lib.aacg_create(C.byref(cfg), C.byref(h))
        if rc:
            raise AacgError(rc, "aacg_create failed (is a GPU visible?)")
        self.handle = h
        self.input_kind = input_kind
        self.max_streams, self.max_channels = max_streams, max_channels

    def close(self):
        if self.handle:
            self.lib.aacg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise AacgError(rc, self.lib.aacg_last_error(self.handle).decode())
        return rc

    # -- host-buffer path -----------------------------------------------------------------
    def decode_batch(self, units, coeffs, meta, n_pcm_floats, tns=None, cce=None, out=None):
        """out: an array to decode into (a caller that reuses its output buffer, as a real host does; default: a fresh one
        filled with NaN so that tests see every sample that was not written)."""
        if cce is not None:
            return self._decode_batch_ex(units, coeffs, meta, n_pcm_floats, tns, cce)
        units = np.ascontiguousarray(units)
        assert units.dtype == UNIT_DTYPE
        coeffs = np.ascontiguousarray(coeffs)
        assert coeffs.dtype == (np.int16 if self.input_kind == INPUT_QUANT_I16 else np.float32)
        n_blocks = coeffs.size // 1024
        if meta is not None:
            meta = np.ascontiguousarray(meta, np.uint16)
        if out is not None:
            assert out.dtype == self.pcm_dtype and out.size >= n_pcm_floats and out.flags["C_CONTIGUOUS"]
            pcm = out
        else:
            pcm = np.full(n_pcm_floats, np.nan, np.float32) if self.pcm_dtype == np.float32 else np.full(n_pcm_floats, -32768, np.int16)
        if tns is not None:
            tns = np.ascontiguousarray(tns)
            assert tns.dtype == TNS_DTYPE
            self._check(self.lib.aacg_decode_batch_tns(self.handle, units.ctypes.data, len(units), coeffs.ctypes.data, n_blocks,
                                                       meta.ctypes.data if meta is not None else None,
                                                       meta.size // 120 if meta is not None else 0,
                                                       tns.ctypes.data, len(tns), pcm.ctypes.data, pcm.size))
            return pcm
        self._check(self.lib.aacg_decode_batch(self.handle, units.ctypes.data, len(units), coeffs.ctypes.data, n_blocks,
                                               meta.ctypes.data if meta is not None else None,
                                               meta.size // 120 if meta is not None else 0, pcm.ctypes.data, pcm.size))
        return pcm

    def _decode_batch_ex(self, units, coeffs, meta, n_pcm_floats, tns, cce):
        """aacg_decode_batch_ex: every array of the batch in one record (coupling side info included)."""
        units, coeffs, cce = np.ascontiguousarray(units), np.ascontiguousarray(coeffs), np.ascontiguousarray(cce)
        assert units.dtype == UNIT_DTYPE and cce.dtype == CCE_DTYPE
        meta = np.ascontiguousarray(meta, np.uint16) if meta is not None else None
        tns = np.ascontiguousarray(tns) if tns is not None else None
        pcm = np.full(n_pcm_floats, np.nan, np.float32)
        b = Batch(units.ctypes.data, len(units), coeffs.ctypes.data, coeffs.size // 1024,
                  meta.ctypes.data if meta is not None else None, meta.size // 120 if meta is not None else 0,
                  tns.ctypes.data if tns is not None else None, len(tns) if tns is not None else 0,
                  cce.ctypes.data, len(cce), pcm.ctypes.data, pcm.size)
        self._check(self.lib.aacg_decode_batch_ex(self.handle, C.byref(b)))
        return pcm

    def submit(self, units, coeffs, meta, pcm, tns=None):
        """Asynchronous decode_batch into the caller's `pcm` array (keep every array alive until wait)."""
        assert units.dtype == UNIT_DTYPE and units.flags.c_contiguous and coeffs.flags.c_contiguous and pcm.flags.c_contiguous
        t = C.c_uint64()
        if tns is not None:
            assert tns.dtype == TNS_DTYPE and tns.flags.c_contiguous
            self._check(self.lib.aacg_submit_tns(self.handle, units.ctypes.data, len(units), coeffs.ctypes.data, coeffs.size // 1024,
                                                 meta.ctypes.data if meta is not None else None,
                                                 meta.size // 120 if meta is not None else 0,
                                                 tns.ctypes.data, len(tns), pcm.ctypes.data, pcm.size, C.byref(t)))
            return t.value
        self._check(self.lib.aacg_submit(self.handle, units.ctypes.data, len(units), coeffs.ctypes.data, coeffs.size // 1024,
                                         meta.ctypes.data if meta is not None else None,
                                         meta.size // 120 if meta is not None else 0, pcm.ctypes.data, pcm.size, C.byref(t)))
        return t.value

    def wait(self, ticket):
        self._check(self.lib.aacg_wait(self.handle, ticket))

    def pinned(self, shape, dtype):
        """numpy array over page-locked host memory (aacg_host_alloc): transfers from/to it are asynchronous."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.aacg_host_alloc(n)
        if not p:
            raise AacgError(-3, "aacg_host_alloc failed")
        buf = (C.c_char * n).from_address(p)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return a

    # -- device-resident path -----------------------------------------------------------------
    def plan(self, units, tns=None, cce=None):
        units = np.ascontiguousarray(units)
        assert units.dtype == UNIT_DTYPE
        h = C.c_void_p()
        if cce is not None:
            cce = np.ascontiguousarray(cce)
            assert cce.dtype == CCE_DTYPE
            tns = np.ascontiguousarray(tns) if tns is not None else None
            self._check(self.lib.aacg_plan_create_ex(self.handle, units.ctypes.data, len(units), tns.ctypes.data if tns is not None else None,
                                                     len(tns) if tns is not None else 0, cce.ctypes.data, len(cce), C.byref(h)))
            return Plan(self, h, len(units))
        if tns is not None:
            tns = np.ascontiguousarray(tns)
            assert tns.dtype == TNS_DTYPE
            self._check(self.lib.aacg_plan_create_tns(self.handle, units.ctypes.data, len(units), tns.ctypes.data, len(tns), C.byref(h)))
            return Plan(self, h, len(units))
        self._check(self.lib.aacg_plan_create(self.handle, units.ctypes.data, len(units), C.byref(h)))
        return Plan(self, h, len(units))

    def decode_device(self, plan, d_coeffs, d_meta, d_pcm, stream=0):
        """d_* are raw device addresses (e.g. torch.Tensor.data_ptr()); stream a hipStream_t handle or 0."""
        self._check(self.lib.aacg_decode_device(self.handle, plan.handle, d_coeffs, d_meta, d_pcm, stream))

    def decode_pipelined(self, plan, d_coeffs, d_meta, d_pcm, mark=None):
        """aacg_decode_pipelined: the next launch of `plan` on the engine's internal streams taken in turn; it may overlap the
        launch before it (their chains meet in rendezvous cells).  Results: after pipeline_join / synchronize.
        mark: a TimerMark bound to the launch's completion (aacg_decode_pipelined_timed; measurement only)."""
        if mark is not None:
            self._check(self.lib.aacg_decode_pipelined_timed(self.handle, plan.handle, d_coeffs, d_meta, d_pcm, mark.h))
        else:
            self._check(self.lib.aacg_decode_pipelined(self.handle, plan.handle, d_coeffs, d_meta, d_pcm))

    def pipeline_fork(self, stream):
        """The pipeline's later launches start after everything enqueued on `stream` so far."""
        self._check(self.lib.aacg_pipeline_fork(self.handle, stream))

    def pipeline_join(self, stream=0):
        """Work enqueued on `stream` from now on starts after the pipeline's launches so far; 0: the host waits for them."""
        self._check(self.lib.aacg_pipeline_join(self.handle, stream or None))

    def pipeline_chained(self):
        """Launches of decode_pipelined that continued (and were allowed to overlap) the launch before them."""
        return int(self.lib.aacg_pipeline_chained(self.handle))

    def pipeline_streams_used(self):
        """How many of the engine's internal streams the current pipelined sequence takes in turn (0 before the first launch)."""
        return int(self.lib.aacg_pipeline_streams_used(self.handle))

    def pipeline_info(self):
        """(streams the current pipelined sequence takes in turn, whether they were seen to run side by side: 1 / 0 / -1 before set-up)."""
        a, b = C.c_int(), C.c_int()
        self._check(self.lib.aacg_pipeline_info(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_wait_limit_ms(self, ms):
        self._check(self.lib.aacg_set_wait_limit_ms(self.handle, int(ms)))

    def debug_set_wait_mode(self, mode, spin_us=-1.0):
        self._check(self.lib.aacg_debug_set_wait_mode(self.handle, mode, float(spin_us)))

    def debug_stall(self, which, ms):
        """One of the engine's streams (0: its own, 1..3: the pipeline's) busy for `ms` milliseconds: tests of the bounded waits."""
        self.lib.aacg_debug_stall.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        self._check(self.lib.aacg_debug_stall(self.handle, which, ms))

    def in_flight(self):
        buf = C.create_string_buffer(4096)
        self.lib.aacg_debug_in_flight(self.handle, buf, 4096)
        return buf.value.decode()

    def pipeline_concurrent(self):
        """True if the engine's internal streams were seen to run side by side (else pipelined launches serialise: correct, not faster)."""
        return bool(self.lib.aacg_pipeline_concurrent(self.handle))

    def plan_refresh_from_parse(self, plan, d_parsed_units, d_results, max_units, d_refused, stream=0):
        """Device pointers: the plan's unit records take what aacg_parse_device wrote (run tables unchanged)."""
        self._check(self.lib.aacg_plan_refresh_from_parse(self.handle, plan.handle, d_parsed_units, d_results, max_units, d_refused, stream))

    def plan_refresh_units(self, plan, units, stream=0):
        """The kept plan takes the next batch's unit records (same structure); raises AacgError(LAYOUT_CHANGE) otherwise."""
        units = np.ascontiguousarray(units)
        assert units.dtype == UNIT_DTYPE
        self._check(self.lib.aacg_plan_refresh_units(self.handle, plan.handle, units.ctypes.data, len(units), stream))

    def spectral_device(self, plan, d_coeffs, d_meta, d_spec, stream=0):
        self._check(self.lib.aacg_spectral_device(self.handle, plan.handle, d_coeffs, d_meta, d_spec, stream))

    def synchronize(self, stream=0):
        self._check(self.lib.aacg_synchronize(self.handle, stream))

    # -- overlap state ------------------------------------------------------------------------
    def get_overlap(self, stream, channel):
        a = np.empty(1024, np.float32)
        self._check(self.lib.aacg_get_overlap(self.handle, stream, channel, a.ctypes.data))
        return a

    def set_overlap(self, stream, channel, values):
        a = np.ascontiguousarray(values, np.float32)
        assert a.size == 1024
        self._check(self.lib.aacg_set_overlap(self.handle, stream, channel, a.ctypes.data))

    def reset_stream(self, stream):
        self._check(self.lib.aacg_reset_stream(self.handle, stream))

    def table(self, which):
        n = self._check(self.lib.aacg_get_table(self.handle, which, np.empty(1, np.float32).ctypes.data, 0))
        a = np.empty(n, np.float32)
        self._check(self.lib.aacg_get_table(self.handle, which, a.ctypes.data, n))
        return a

    def kernel_name(self):
        return self.lib.aacg_kernel_name().decode()

    def debug_set_route(self, flags):
        """Diagnostic route choices for parity tests (aacg_debug_set_route): 1 = independent coupling as the separate pass."""
        self._check(self.lib.aacg_debug_set_route(self.handle, flags))

    def plan_kernels(self, plan, pipelined=False):
        """The launches aacg_decode_device (pipelined: aacg_decode_pipelined) makes for this plan, by kernel name (what a rocprofv3 kernel trace shows)."""
        buf = C.create_string_buffer(512)
        self._check(self.lib.aacg_plan_kernels_ex(self.handle, plan.handle, 1 if pipelined else 0, buf, 512))
        return buf.value.decode()
