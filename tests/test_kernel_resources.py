"""Register / scratch / LDS budget of the hot kernels, from hipcc's own resource report (-Rpass-analysis=kernel-resource-usage),
in the build container: a toolchain change (or an edit) that spills, drops occupancy or outgrows the CU's LDS fails HERE, next
to tests/test_disasm_guard.py, instead of showing up as a slower or failing launch on the GPU box.

The budgets are what the design rests on: the 16-wave run kernels hold one workgroup per CU — 4 waves per SIMD,
so at most 128 VGPRs, no scratch (a scratch reload waits for the PCM stores in flight), at most 160 KiB of LDS."""
import concurrent.futures
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aac.js_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
SCHED = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
# translation unit -> (extra flags as the Makefile builds it, {kernel: (max VGPRs, min waves/SIMD, max LDS bytes)})
WIDE = (128, 4, 160 * 1024)
TUS = {
    "aacg_engine_rv.hip": (SCHED, {"aacg_imdct_run_quant_rv": WIDE, "aacg_imdct_run_f32_rv": WIDE, "aacg_imdct_run_quant_rv_nt": WIDE, "aacg_imdct_run_f32_rv_nt": WIDE}),
    "aacg_engine_nt.hip": (SCHED, {"aacg_imdct_run_quant_nt": WIDE, "aacg_imdct_run_f32_nt": WIDE}),
    "aacg_engine.hip": (SCHED, {"aacg_imdct_run_quant": WIDE, "aacg_imdct_run_f32": WIDE}),
    "aacg_engine_ext.hip": (SCHED, {"aacg_imdct_run_quant_dd": WIDE, "aacg_imdct_run_f32_dd": WIDE}),
    "aacg_engine_i16.hip": (SCHED, {"aacg_imdct_run_quant_i16": WIDE, "aacg_imdct_run_f32_i16": WIDE, "aacg_imdct_run_quant_i16_nt": WIDE, "aacg_imdct_run_f32_i16_nt": WIDE}),
    "aacg_engine_exrun.hip": (SCHED, {"aacg_imdct_run_quant_ex": WIDE, "aacg_imdct_run_f32_ex": WIDE}),
    "aacg_engine_couple.hip": ([], {"aacg_imdct_run_quant_cpl": WIDE, "aacg_imdct_run_f32_cpl": WIDE, "aacg_imdct_run_quant_cpl_nt": WIDE, "aacg_imdct_run_f32_cpl_nt": WIDE}),
}


def resource_report(tu, flags):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function"] + flags +
                       ["-Rpass-analysis=kernel-resource-usage", "-c", tu, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?): (\S+) \[-Rpass", line)
        if m and name:
            out[name][m.group(1).strip()] = m.group(2)
    return out


@pytest.fixture(scope="module")
def reports():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        futs = {tu: ex.submit(resource_report, tu, flags) for tu, (flags, _) in TUS.items()}
        return {tu: f.result() for tu, f in futs.items()}


@pytest.mark.parametrize("tu", sorted(TUS))
def test_hot_kernels_keep_their_register_and_lds_budget(reports, tu):
    for kernel, (max_vgpr, min_occ, max_lds) in TUS[tu][1].items():
        rep = reports[tu].get(kernel)
        assert rep, "%s: kernel %s not in the resource report (%s)" % (tu, kernel, sorted(reports[tu]))
        assert int(rep["VGPRs"]) + int(rep["AGPRs"]) <= max_vgpr, (kernel, rep)
        assert int(rep["ScratchSize [bytes/lane]"]) == 0 and int(rep["VGPRs Spill"]) == 0, (kernel, rep)
        # scalar spills go to VGPR lanes (no memory traffic); the plain kernels have none, the optional-stage builds a handful
        assert int(rep["SGPRs Spill"]) <= (32 if kernel.endswith("_ex") else 0), (kernel, rep)
        assert int(rep["Occupancy [waves/SIMD]"]) >= min_occ, (kernel, rep)
        assert int(rep["LDS Size [bytes/block]"]) <= max_lds, (kernel, rep)
        assert rep["Dynamic Stack"] == "False", (kernel, rep)
