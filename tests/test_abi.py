"""The C-ABI shared library loads on a machine without a GPU and exports every symbol
include/aacgpu.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

import aacgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="aacgpu.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aacg_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(aacgpu.ABI_SYMBOLS)
    assert declared_symbols("aacgpu_tools.h") == sorted(aacgpu.TOOLS_SYMBOLS)
    # a host needs aacgpu.h only: no measurement / diagnostic entry point is declared there
    assert not [n for n in declared_symbols() if n.startswith(("aacg_debug_", "aacg_timer_", "aacg_calib_"))]


def test_library_exports_every_declared_symbol(engine_lib):
    for name in declared_symbols() + declared_symbols("aacgpu_tools.h"):
        assert hasattr(engine_lib, name), "libaacgpu.so does not export " + name
    assert engine_lib.aacg_abi_version() == 6
    assert engine_lib.aacg_kernel_name().decode().startswith("aacg_imdct_run")


def test_unit_desc_layout_matches_header():
    """aacg_unit_desc is 64 bytes, aacg_band_meta 240, as the JS host packs them."""
    src = r'''
    #include "include/aacgpu.h"
    #include <stddef.h>
    int sizes[] = { sizeof(aacg_unit_desc), sizeof(aacg_chan_info), sizeof(aacg_band_meta), sizeof(aacg_config),
                    offsetof(aacg_unit_desc, coef_offset), offsetof(aacg_unit_desc, ch), offsetof(aacg_chan_info, group_len) };
    '''
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        so = os.path.join(d, "s.so")
        subprocess.run(["gcc", "-shared", "-fPIC", "-I", ROOT, "-o", so, os.path.join(d, "s.c")], check=True)
        arr = (ctypes.c_int * 7).in_dll(ctypes.CDLL(so), "sizes")
        assert list(arr) == [64, 16, 240, 44, 16, 24, 8]
    assert aacgpu.UNIT_DTYPE.itemsize == 64
    assert aacgpu.UNIT_DTYPE.fields["coef_offset"][1] == 16 and aacgpu.UNIT_DTYPE.fields["ch"][1] == 24


def test_parser_record_layouts_match_header():
    """aacg_code_entry / aacg_parse_frame / aacg_parse_result as the Python and JavaScript bindings pack them."""
    src = r'''
    #include "include/aacgpu.h"
    #include <stddef.h>
    int sizes[] = { sizeof(aacg_code_entry), offsetof(aacg_code_entry, len), offsetof(aacg_code_entry, v), sizeof(aacg_parse_frame),
                    sizeof(aacg_parse_result), offsetof(aacg_parse_result, bits_used), sizeof(aacg_tns_info) };
    '''
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        so = os.path.join(d, "s.so")
        subprocess.run(["gcc", "-shared", "-fPIC", "-I", ROOT, "-o", so, os.path.join(d, "s.c")], check=True)
        arr = (ctypes.c_int * 7).in_dll(ctypes.CDLL(so), "sizes")
        assert list(arr) == [12, 4, 5, 8, 8, 4, 424]
    assert aacgpu.CODE_ENTRY_DTYPE.itemsize == 12 and aacgpu.CODE_ENTRY_DTYPE.fields["v"][1] == 5
    assert aacgpu.PARSE_FRAME_DTYPE.itemsize == 8 and aacgpu.PARSE_RESULT_DTYPE.itemsize == 8


def test_parser_create_without_gpu_fails_loudly(engine_lib):
    """The device front end has no CPU path either."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import aacgpu, aacgpu_workload\n"
            "e, c = aacgpu_workload.standin_codebooks()\n"
            "try:\n    aacgpu.Parser(e, c)\n    print('created')\n"
            "except aacgpu.AacgError as x:\n    print('refused', x.code)\n") % os.path.join(ROOT, "aac.js_amd", "python")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert "refused -2" in r.stdout, r.stdout + r.stderr


def test_create_without_gpu_fails_loudly(engine_lib):
    """No silent CPU path: without a device aacg_create returns an error."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import aacgpu\n"
            "try:\n    aacgpu.Engine()\n    print('created')\n"
            "except aacgpu.AacgError as e:\n    print('refused', e.code)\n") % os.path.join(ROOT, "aac.js_amd", "python")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert "refused -2" in r.stdout, r.stdout + r.stderr
