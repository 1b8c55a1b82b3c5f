/*
 * aacg_engine_exrun.hip — the run kernels with the optional stages inside (imdct_run_body<..., EX = true>): noise bands
 * (AACG_PNS_SPEC) and TNS filters (AACG_TNS_SPEC) applied between dequantisation and IMDCT, in the wave's own slot.
 * Batches that carry TNS records or noise bands, produce f32 PCM, have no coupling elements and no chain longer than a
 * run's double-duty variant is needed for (with the filters in both copies of its front end that variant spilled 800
 * bytes per lane) run here in ONE launch; everything else with optional stages takes the staged route
 * (aacg_engine_spectral.hip).  A code object of its own,
 * so that the plain kernels of aacg_engine.hip never move.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_ex(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, true>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_ex(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, true>(P); }

/* the same with chains that meet in rendezvous cells (aacg_engine_rv.hip): chains longer than a run without a staged route, and
 * launches through the pipeline that overlap */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_ex_rv(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, true, false, true>(P, &V); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_ex_rv(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, true, false, true>(P, &V); }

static_assert(AACG_LDS_BYTES_QUANT_EX <= 160 * 1024, "the TNS exchange areas must fit beside the slots");

const aacg_run_kernel aacg_run_kernels_exrun[] = {
    {AACG_RK_EX | AACG_RK_QUANT, "aacg_imdct_run_quant_ex", (const void*)aacg_imdct_run_quant_ex},
    {AACG_RK_EX, "aacg_imdct_run_f32_ex", (const void*)aacg_imdct_run_f32_ex},
    {AACG_RK_EX | AACG_RK_RV | AACG_RK_QUANT, "aacg_imdct_run_quant_ex_rv", (const void*)aacg_imdct_run_quant_ex_rv},
    {AACG_RK_EX | AACG_RK_RV, "aacg_imdct_run_f32_ex_rv", (const void*)aacg_imdct_run_f32_ex_rv}
};
const int aacg_run_kernels_exrun_n = 4;
