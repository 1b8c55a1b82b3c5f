/*
 * aacg_engine_half.hip — the rendezvous run kernels as workgroups of EIGHT waves (imdct_run_body<..., NW = 8>): runs of up to eight
 * frames, 64 KB of slots and a table block without the windows (read from global memory), < 80 KB of LDS per workgroup — so that
 * TWO workgroups share a CU and one of them computes while the other waits for its first spectra or for its last stores to
 * land.  Same waves, same 128 registers, same arithmetic as aacg_engine_rv.hip; the chains' extra cuts are rendezvous cells like
 * any other.  Their own translation unit and code object.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

#define AACG_HALF_THREADS (AACG_HALF_WAVES * 64)
static_assert(AACG_LDS_BYTES_HALF_QUANT <= 80 * 1024 && AACG_LDS_BYTES_HALF_F32 <= 80 * 1024, "two workgroups per CU");

/* (512 threads, 4 waves per SIMD): two such workgroups fill a CU's sixteen wave slots at 128 registers each */
extern "C" __global__ __launch_bounds__(AACG_HALF_THREADS, 4)
void aacg_imdct_run_quant_rv_h(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true, false, AACG_HALF_WAVES>(P, &V); }

extern "C" __global__ __launch_bounds__(AACG_HALF_THREADS, 4)
void aacg_imdct_run_f32_rv_h(const aacg_kparams P, const aacg_rv_args V) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true, false, AACG_HALF_WAVES>(P, &V); }

const aacg_run_kernel aacg_run_kernels_half[] = {
    {AACG_RK_RV | AACG_RK_HALF | AACG_RK_QUANT, "aacg_imdct_run_quant_rv_h", (const void*)aacg_imdct_run_quant_rv_h},
    {AACG_RK_RV | AACG_RK_HALF, "aacg_imdct_run_f32_rv_h", (const void*)aacg_imdct_run_f32_rv_h}
};
const int aacg_run_kernels_half_n = 2;
