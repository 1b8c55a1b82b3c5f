/*
 * aacg_engine_i16.hip — the run kernels for AACG_OUTPUT_I16 engines: the same bodies with the epilogue's stores
 * narrowed to int16 (round to nearest, saturating).  Their own translation unit, like the other variants, so that
 * the float32 kernels' code objects do not move.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_i16(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_i16(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_dd_i16(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16, true>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_dd_i16(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16, true>(P); }

/* batches of multichannel frames: non-temporal loads of the spectra (aacg_engine_nt.hip says why) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_i16_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16, false, false, false, false, true>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_i16_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16, false, false, false, false, true>(P); }

void aacg_i16_launch(bool quant, bool dd, bool wide, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P)
{
    if (wide && !dd) {
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_i16_nt, grid, block, 0, s, P);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32_i16_nt, grid, block, 0, s, P);
    } else if (dd) {
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_dd_i16, grid, block, 0, s, P);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32_dd_i16, grid, block, 0, s, P);
    } else {
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_i16, grid, block, 0, s, P);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32_i16, grid, block, 0, s, P);
    }
}
