/*
 * aacg_engine_i16.hip — the run kernels for AACG_OUTPUT_I16 engines: the same bodies with the epilogue's stores
 * narrowed to int16 (round to nearest, saturating).  Their own translation unit, like the other variants, so that
 * the float32 kernels' code objects do not move.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

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

/* chains that meet in rendezvous cells — between the runs of a launch and between consecutive launches (aacg_engine_rv.hip) */
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_rv_i16(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16, false, false, false, true>(P, &V); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_rv_i16(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16, false, false, false, true>(P, &V); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_rv_i16_nt(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_I16, false, false, false, true, true>(P, &V); }
extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_rv_i16_nt(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_I16, false, false, false, true, true>(P, &V); }

const aacg_run_kernel aacg_run_kernels_i16[] = {
    {AACG_RK_RV | AACG_RK_I16 | AACG_RK_QUANT, "aacg_imdct_run_quant_rv_i16", (const void*)aacg_imdct_run_quant_rv_i16},
    {AACG_RK_RV | AACG_RK_I16, "aacg_imdct_run_f32_rv_i16", (const void*)aacg_imdct_run_f32_rv_i16},
    {AACG_RK_RV | AACG_RK_I16 | AACG_RK_NT | AACG_RK_QUANT, "aacg_imdct_run_quant_rv_i16_nt", (const void*)aacg_imdct_run_quant_rv_i16_nt},
    {AACG_RK_RV | AACG_RK_I16 | AACG_RK_NT, "aacg_imdct_run_f32_rv_i16_nt", (const void*)aacg_imdct_run_f32_rv_i16_nt},
    {AACG_RK_I16 | AACG_RK_QUANT, "aacg_imdct_run_quant_i16", (const void*)aacg_imdct_run_quant_i16},
    {AACG_RK_I16, "aacg_imdct_run_f32_i16", (const void*)aacg_imdct_run_f32_i16},
    {AACG_RK_I16 | AACG_RK_DD | AACG_RK_QUANT, "aacg_imdct_run_quant_dd_i16", (const void*)aacg_imdct_run_quant_dd_i16},
    {AACG_RK_I16 | AACG_RK_DD, "aacg_imdct_run_f32_dd_i16", (const void*)aacg_imdct_run_f32_dd_i16},
    {AACG_RK_I16 | AACG_RK_NT | AACG_RK_QUANT, "aacg_imdct_run_quant_i16_nt", (const void*)aacg_imdct_run_quant_i16_nt},
    {AACG_RK_I16 | AACG_RK_NT, "aacg_imdct_run_f32_i16_nt", (const void*)aacg_imdct_run_f32_i16_nt}
};
const int aacg_run_kernels_i16_n = 10;
