/*
 * aacg_engine_tns.hip — the run kernel with the TNS stage compiled in (AACG_TNS_SPEC batches that carry TNS
 * side info; the planner gives those no full later runs).  Its own translation unit: compiled with LLVM's
 * default machine scheduler, which suits the long dependent chains of tns_pass better than the ILP-first one
 * the other kernels use (94.7 vs 100.8 us on config 3 with TNS).  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_tns(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, true>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_tns(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, true>(P); }

int aacg_tns_set_lds_limits(void)
{
    hipError_t rc = hipFuncSetAttribute((const void*)aacg_imdct_run_quant_tns, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_QUANT);
    if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)aacg_imdct_run_f32_tns, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_F32);
    return rc == hipSuccess ? 0 : -1;
}

void aacg_tns_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P)
{
    if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_tns, grid, block, AACG_LDS_BYTES_QUANT, s, P);
    else       hipLaunchKernelGGL(aacg_imdct_run_f32_tns, grid, block, AACG_LDS_BYTES_F32, s, P);
}
