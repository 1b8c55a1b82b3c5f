/*
 * aacg_engine_stream.hip — the run kernel for stream-resident runs (multichannel streams, aacg_sr_run): one workgroup
 * walks a stream's frames with all elements of a frame side by side, PCM interleaved through an LDS staging area and
 * stored as full lines (the interleave of decoder.js:203-215 for C > 2).  Its own translation unit, like the other
 * variants.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#define DP_LANE_OPAQUE      /* devport.h: the frame loop must not hoist lane-derived addresses (1.2 KB of spills per lane) */
#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_stream_quant(const aacg_kparams P) { imdct_stream_body<AACG_INPUT_QUANT_I16>(P); }

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_imdct_stream_f32(const aacg_kparams P) { imdct_stream_body<AACG_INPUT_SPEC_F32>(P); }

#define AACG_LDS_BYTES_MAX (160 * 1024)

int aacg_sr_set_lds_limits(void)
{
    hipError_t rc = hipFuncSetAttribute((const void*)aacg_imdct_stream_quant, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_MAX);
    if (rc == hipSuccess) rc = hipFuncSetAttribute((const void*)aacg_imdct_stream_f32, hipFuncAttributeMaxDynamicSharedMemorySize, AACG_LDS_BYTES_MAX);
    return rc == hipSuccess ? 0 : -1;
}

/* lds_floats: slots + staging + counters of the plan's largest run (aacg_plan_host.sr_lds_floats); the tables come on top */
int aacg_sr_launch(bool quant, unsigned n_runs, unsigned lds_floats, hipStream_t s, const aacg_kparams& P)
{
    const size_t bytes = ((size_t)(quant ? AACG_TAB_QUANT_FLOATS : AACG_TAB_F32_FLOATS) + lds_floats) * sizeof(float);
    if (bytes > AACG_LDS_BYTES_MAX) return -1;
    if (quant) hipLaunchKernelGGL(aacg_imdct_stream_quant, dim3(n_runs), dim3(AACG_WG_THREADS), bytes, s, P);
    else       hipLaunchKernelGGL(aacg_imdct_stream_f32, dim3(n_runs), dim3(AACG_WG_THREADS), bytes, s, P);
    return 0;
}
