/*
 * aacg_engine_ext.hip — the run kernel's variants (see aacg_engine.hip): separate translation unit, separate
 * code object.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"


/* Variants of aacg_imdct_run_quant / _f32:
 *   _dd : plans with full later runs (chains longer than 16 frames), whose first wave does double duty;
 * (the optional TNS / PNS stages are a kernel of their own, aacg_engine_spectral.hip) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_dd(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, true>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_dd(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, true>(P); }



const aacg_run_kernel aacg_run_kernels_ext[] = {
    {AACG_RK_DD | AACG_RK_QUANT, "aacg_imdct_run_quant_dd", (const void*)aacg_imdct_run_quant_dd},
    {AACG_RK_DD, "aacg_imdct_run_f32_dd", (const void*)aacg_imdct_run_f32_dd}
};
const int aacg_run_kernels_ext_n = 2;
