/*
 * aacg_engine_rv.hip — the 16-wave run kernels for plans whose chains are longer than a run, WITHOUT a recomputed frame: the
 * runs of a chain hand their tails over through a rendezvous cell in global memory (aacg_rv_args; imdct_run_body<..., RV = true>).
 * Their own translation unit and code object, like the other variants.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_rv(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true>(P, &V); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_rv(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true>(P, &V); }

/* the same for batches of multichannel frames: non-temporal loads of the spectra (aacg_engine_nt.hip says why) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_rv_nt(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true, true>(P, &V); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_rv_nt(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true, true>(P, &V); }

void aacg_rv_launch(bool quant, bool wide, dim3 grid, dim3 block, hipStream_t s, const aacg_kparams& P, const aacg_rv_args& V)
{
    if (wide) {
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_rv_nt, grid, block, 0, s, P, V);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32_rv_nt, grid, block, 0, s, P, V);
    } else {
        if (quant) hipLaunchKernelGGL(aacg_imdct_run_quant_rv, grid, block, 0, s, P, V);
        else       hipLaunchKernelGGL(aacg_imdct_run_f32_rv, grid, block, 0, s, P, V);
    }
}
