/*
 * aacg_engine_couple.hip — AACG_CCE_SPEC: the coupling launches (aacg_kernels.h: couple_spec_body, couple_pcm_body).
 * Small element-wise kernels between the stages of a batch that carries coupling channel elements; batches without
 * them never come here.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

#define AACG_COUPLE_WAVES 4

extern "C" __global__ __launch_bounds__(AACG_COUPLE_WAVES * 64)
void aacg_couple_spec(const aacg_couple_params Q) { couple_spec_body(Q, AACG_COUPLE_WAVES); }

extern "C" __global__ __launch_bounds__(AACG_COUPLE_WAVES * 64)
void aacg_couple_pcm(const aacg_couple_params Q) { couple_pcm_body(Q, AACG_COUPLE_WAVES); }

/* the run kernels with the independent coupling in their epilogue (plans whose every run is a first run or a run with a
 * wave of its own for the predecessor: no double duty) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_cpl(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, true>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_cpl(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, true>(P); }

/* batches of multichannel frames (where coupling lives): non-temporal loads of the spectra (aacg_engine_nt.hip says why) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_cpl_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, true, false, true>(P); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_cpl_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, true, false, true>(P); }

const aacg_run_kernel aacg_run_kernels_couple[] = {
    {AACG_RK_CPL | AACG_RK_QUANT, "aacg_imdct_run_quant_cpl", (const void*)aacg_imdct_run_quant_cpl},
    {AACG_RK_CPL, "aacg_imdct_run_f32_cpl", (const void*)aacg_imdct_run_f32_cpl},
    {AACG_RK_CPL | AACG_RK_NT | AACG_RK_QUANT, "aacg_imdct_run_quant_cpl_nt", (const void*)aacg_imdct_run_quant_cpl_nt},
    {AACG_RK_CPL | AACG_RK_NT, "aacg_imdct_run_f32_cpl_nt", (const void*)aacg_imdct_run_f32_cpl_nt}
};
const int aacg_run_kernels_couple_n = 4;

void aacg_couple_launch(bool pcm, hipStream_t s, const aacg_couple_params& Q)
{
    if (Q.n_jobs <= 0) return;
    const dim3 grid((unsigned)((Q.n_jobs + AACG_COUPLE_WAVES - 1) / AACG_COUPLE_WAVES)), block(AACG_COUPLE_WAVES * 64);
    if (pcm) hipLaunchKernelGGL(aacg_couple_pcm, grid, block, 0, s, Q);
    else     hipLaunchKernelGGL(aacg_couple_spec, grid, block, 0, s, Q);
}
