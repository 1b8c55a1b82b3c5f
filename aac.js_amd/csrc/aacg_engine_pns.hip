/*
 * aacg_engine_pns.hip — AACG_PNS_SPEC: the spectral stage (dequantisation, mid/side, intensity) followed by the
 * noise bands, as a kernel of its own that writes f32 spectra; the f32 run kernel takes them from there.  Batches
 * without NOISE_BT bands never come here.  MI355X (gfx950) only.
 */
#include <hip/hip_runtime.h>

#include "aacg_kernels.h"

extern "C" __global__ __launch_bounds__(AACG_WG_THREADS)
void aacg_spectral_pns(const aacg_kparams P, int n_units) { spectral_pns_body(P, n_units); }

#define AACG_LDS_BYTES_SPECTRAL_PNS ((AACG_TAB_QUANT_FLOATS + AACG_WG_WAVES * 1024) * 4)

int aacg_pns_set_lds_limits(void)
{
    return hipFuncSetAttribute((const void*)aacg_spectral_pns, hipFuncAttributeMaxDynamicSharedMemorySize,
                               AACG_LDS_BYTES_SPECTRAL_PNS) == hipSuccess ? 0 : -1;
}

void aacg_pns_launch(int n_units, hipStream_t s, const aacg_kparams& P)
{
    hipLaunchKernelGGL(aacg_spectral_pns, dim3((unsigned)((n_units + AACG_WG_WAVES - 1) / AACG_WG_WAVES)), dim3(AACG_WG_THREADS),
                       AACG_LDS_BYTES_SPECTRAL_PNS, s, P, n_units);
}
