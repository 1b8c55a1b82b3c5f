/*
 * aacg_engine_couple.hip — AACG_CCE_SPEC: the coupling launches (aacg_kernels.h: couple_spec_body, couple_pcm_body).
 * Small element-wise kernels between the stages of a batch that carries coupling channel elements; batches without
 * them never come here.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

#define AACG_COUPLE_WAVES 4

extern "C" __global__ __launch_bounds__(AACG_COUPLE_WAVES * 64)
void aacg_couple_spec(const aacg_couple_params Q) { couple_spec_body(Q, AACG_COUPLE_WAVES); }

extern "C" __global__ __launch_bounds__(AACG_COUPLE_WAVES * 64)
void aacg_couple_pcm(const aacg_couple_params Q) { couple_pcm_body(Q, AACG_COUPLE_WAVES); }

void aacg_couple_launch(bool pcm, hipStream_t s, const aacg_couple_params& Q)
{
    if (Q.n_jobs <= 0) return;
    const dim3 grid((unsigned)((Q.n_jobs + AACG_COUPLE_WAVES - 1) / AACG_COUPLE_WAVES)), block(AACG_COUPLE_WAVES * 64);
    if (pcm) hipLaunchKernelGGL(aacg_couple_pcm, grid, block, 0, s, Q);
    else     hipLaunchKernelGGL(aacg_couple_spec, grid, block, 0, s, Q);
}
