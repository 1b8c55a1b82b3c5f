/*
 * aacg_engine_rv.hip — the 16-wave run kernels for plans whose chains are longer than a run, WITHOUT a recomputed frame: the
 * runs of a chain hand their tails over through a rendezvous cell in global memory (aacg_rv_args; imdct_run_body<..., RV = true>).
 * Their own translation unit and code object, like the other variants.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"
#include "aacg_routes.h"

/* (imdct_run_body<KIND, OUT, DD, EX, CPL, RV, NTL, PRE>) */
AACG_RUN_KERNEL_PRE(aacg_imdct_run_quant_rv, AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true, false)
AACG_RUN_KERNEL_PRE(aacg_imdct_run_f32_rv, AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true, false)
/* the same for batches of multichannel frames: non-temporal loads of the spectra (aacg_engine_nt.hip says why) */
AACG_RUN_KERNEL_PRE(aacg_imdct_run_quant_rv_nt, AACG_INPUT_QUANT_I16, AACG_OUTPUT_F32, false, false, false, true, true)
AACG_RUN_KERNEL_PRE(aacg_imdct_run_f32_rv_nt, AACG_INPUT_SPEC_F32, AACG_OUTPUT_F32, false, false, false, true, true)

const aacg_run_kernel aacg_run_kernels_rv[] = {
    {AACG_RK_RV | AACG_RK_QUANT, "aacg_imdct_run_quant_rv", (const void*)aacg_imdct_run_quant_rv, true},
    {AACG_RK_RV, "aacg_imdct_run_f32_rv", (const void*)aacg_imdct_run_f32_rv, true},
    {AACG_RK_RV | AACG_RK_NT | AACG_RK_QUANT, "aacg_imdct_run_quant_rv_nt", (const void*)aacg_imdct_run_quant_rv_nt, true},
    {AACG_RK_RV | AACG_RK_NT, "aacg_imdct_run_f32_rv_nt", (const void*)aacg_imdct_run_f32_rv_nt, true}
};
const int aacg_run_kernels_rv_n = 4;
