/*
 * aacg_engine_spectral.hip — the optional stages as a kernel of their own: noise bands (AACG_PNS_SPEC) and the TNS
 * filters (AACG_TNS_SPEC), for quantised or f32 input, writing f32 spectra that the f32 run kernel then consumes.
 * Batches without noise bands and without TNS side info never come here.  Its own translation unit, compiled with
 * LLVM's default machine scheduler (it suits the long dependent chains of the TNS passes).  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_spectral_ex_quant(const aacg_kparams P, int n_units) { spectral_ex_body<AACG_INPUT_QUANT_I16>(P, n_units); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_spectral_ex_f32(const aacg_kparams P, int n_units) { spectral_ex_body<AACG_INPUT_SPEC_F32>(P, n_units); }

/* The transition matrices of a plan's TNS filters (tns_matrix_row, aacg_kernels.h): one wave per channel record, one lane per
 * row of each of its three long-window filter slots.  Launched once, when the records are uploaded. */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_tns_matrices(const aacg_dev_tns* recs, double* M, uint32_t n_records) { tns_matrices_body(recs, M, n_records); }

void aacg_tns_matrices_launch(const aacg_dev_tns* d_recs, double* d_m, uint32_t n_records, hipStream_t s)
{
    if (!n_records) return;
    hipLaunchKernelGGL(aacg_tns_matrices, dim3((n_records + AACG_WG_WAVES - 1) / AACG_WG_WAVES), dim3(AACG_WG_THREADS), 0, s, d_recs, d_m, n_records);
}

#define AACG_LDS_BYTES_EX_QUANT ((AACG_SPX_TAB_FLOATS + AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4)
#define AACG_LDS_BYTES_EX_F32   ((AACG_WG_WAVES * AACG_SPX_WAVE_FLOATS) * 4)

int aacg_spectral_ex_set_lds_limits(void)
{
    hipError_t rc = hipFuncSetAttribute((const void*)aacg_spectral_ex_quant, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_EX_QUANT);
    if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)aacg_spectral_ex_f32, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_EX_F32);
    return rc == hipSuccess ? 0 : -1;
}

void aacg_spectral_ex_launch(bool quant, int n_units, hipStream_t s, const aacg_kparams& P)
{
    const dim3 grid((unsigned)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES)), block(AACG_WG_THREADS);
    if (quant) hipLaunchKernelGGL(aacg_spectral_ex_quant, grid, block, AACG_LDS_BYTES_EX_QUANT, s, P, n_units);
    else       hipLaunchKernelGGL(aacg_spectral_ex_f32, grid, block, AACG_LDS_BYTES_EX_F32, s, P, n_units);
}
