/*
 * aacg_engine_nt.hip — the run kernels for multichannel batches (see aacg_engine.hip): separate translation unit, separate
 * code object.  MI355X (gfx950) only.
 *
 * A frame of more than two channels leaves the kernel element by element: a channel pair writes 8 bytes of every 4 C, and the
 * L2 puts the elements' pieces together — millions of small write requests per launch, which is what bounds these batches
 * (DESIGN.md section 8, config 5).  These variants load their spectra with the non-temporal hint (dp_load_nt): the same HBM
 * traffic, but 80 MB of read-once input per launch no longer allocate lines in that L2; and they skip the per-frame issue
 * priorities.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_quant_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, false, true>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_run_f32_nt(const aacg_kparams P) { imdct_run_body<AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, false, true>(P); }

const aacg_run_kernel aacg_run_kernels_nt[] = {
    {AACG_RK_NT | AACG_RK_QUANT, "aacg_imdct_run_quant_nt", (const void*)aacg_imdct_run_quant_nt},
    {AACG_RK_NT, "aacg_imdct_run_f32_nt", (const void*)aacg_imdct_run_f32_nt}
};
const int aacg_run_kernels_nt_n = 2;
