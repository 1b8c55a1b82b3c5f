/*
 * aacg_engine8.hip — the run kernels at 8 waves per SIMD (aacg_kernels8.h): one channel per wave, <= 64 VGPRs, 80 KB of
 * LDS per 16-wave workgroup, two workgroups per CU.  Their own translation unit and code object.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels8.h"

/* 1024 threads, 8 waves per SIMD: the register allocator is held to 64 VGPRs */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS, 8)
void aacg_imdct_run8_quant(const aacg_kparams8 P) { imdct_run8_body<AACG_INPUT_QUANT_I16>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS, 8)
void aacg_imdct_run8_f32(const aacg_kparams8 P) { imdct_run8_body<AACG_INPUT_SPEC_F32>(P); }

void aacg_run8_launch(bool quant, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams8& P)
{
    if (quant) hipLaunchKernelGGL(aacg_imdct_run8_quant, grid, block, 0, s, P);
    else       hipLaunchKernelGGL(aacg_imdct_run8_f32, grid, block, 0, s, P);
}
