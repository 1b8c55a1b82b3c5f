/*
 * aacg_engine_debug.hip — aacg_debug_transform (include/aacgpu.h): the IMDCT stage of the run kernels on its own, for
 * known-answer tests against the reference's MDCT / FFT vectors (mdct.js:62-115, fft.js:105-192).  One wave runs
 * long_channels<1> or short_channels<1> — the very functions the run kernels call — on one spectrum with a table set
 * whose windows are 1 (so the window stage multiplies by one and the raw IMDCT output comes out) and, optionally, whose
 * pre / post rotation is the identity (what is left is the N/4-point complex inverse FFT in the MDCT's output order).
 * A diagnostic entry point: nothing on the decode path calls it.
 */
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "aacg_kernels.h"
#include "aacg_host.h"

extern "C" __global__ __launch_bounds__(64)
void aacg_debug_transform_kernel(const aacg_tables* tab_global, const float* in, float* out, int is_short)
{
    float* lds = (float*)dp_lds();
    float* area = lds + AACG_TAB_F32_FLOATS;
    const int lane = dp_lane();
    for (int i = lane; i < AACG_TAB_F32_FLOATS; i += 64) lds[i] = ((const float*)tab_global)[i];
    for (int i = lane; i < 1024; i += 64) area[i] = in[i];
    dp_wave_sync();
    chan_par cp[1];
    cp[0].seq = (is_short & 1) ? AACG_EIGHT_SHORT_SEQUENCE : AACG_ONLY_LONG_SEQUENCE; cp[0].shape = 0; cp[0].shape_prev = 0;
    float* const areas[1] = {area};
    float hx[1][8], hy[1][8];
    const bool vm = (is_short & 2) != 0;               /* the int16 seam's variants: mirror-lane exchanges as DPP, columns dealt out by long_col */
    if (is_short & 1) {
        if (vm) short_channels<1, true>(lds, cp, areas, hx, hy); else short_channels<1>(lds, cp, areas, hx, hy);
        const int w = lane >> 3, g = lane & 7;
#pragma unroll
        for (int m = 0; m < 8; m++) { out[128 * w + 2 * g + 16 * m] = hx[0][m]; out[128 * w + 2 * g + 16 * m + 1] = hy[0][m]; }
    } else {
        if (vm) long_channels<1, true>(lds, cp, true, areas, hx, hy); else long_channels<1>(lds, cp, true, areas, hx, hy);
        dp_wave_sync();
        const int col = vm ? long_col(lane) : lane;
#pragma unroll
        for (int m = 0; m < 8; m++) { out[2 * col + 128 * m] = hx[0][m]; out[2 * col + 128 * m + 1] = hy[0][m]; }
        for (int i = lane; i < 1024; i += 64) out[1024 + i] = area[i];
    }
}

/* in: 1024 floats (one long spectrum, or 8 short windows of 128); out: 2048 floats (long: the IMDCT output y[0..2047];
 * short: s[0..1023] with s[128 w + i] = y_(w-1)[128 + i] + y_w[i], the second 1024 floats untouched). */
extern "C" int aacg_debug_transform(int device_ordinal, int sample_index, int is_short, int identity_rotation, const float* in, float* out)
{
    if (!in || !out) return AACG_ERR_INVALID_ARG;
    std::vector<aacg_tables> t(1);
    if (aacg_build_tables(sample_index, t.data(), nullptr) != AACG_OK) return AACG_ERR_INVALID_ARG;
    for (int s = 0; s < 2; s++) {
        for (int i = 0; i < 1024; i++) t[0].win_long[s][i] = 1.0f;
        for (int i = 0; i < 128; i++) t[0].win_short[s][i] = 1.0f;
    }
    if (identity_rotation) {
        for (int j = 0; j < 8; j++) for (int l = 0; l < 64; l++) { t[0].sincos_long[j][l].re = 1.0f; t[0].sincos_long[j][l].im = 0.0f; }
        for (int j = 0; j < 8; j++) for (int g = 0; g < 8; g++) { t[0].sincos_short[j][g].re = 1.0f; t[0].sincos_short[j][g].im = 0.0f; }
    }
    if (hipSetDevice(device_ordinal) != hipSuccess) return AACG_ERR_NO_DEVICE;
    aacg_tables* d_tab = nullptr; float* d_in = nullptr; float* d_out = nullptr;
    int rc = AACG_OK;
    if (hipMalloc((void**)&d_tab, sizeof(aacg_tables)) != hipSuccess || hipMalloc((void**)&d_in, 4096) != hipSuccess || hipMalloc((void**)&d_out, 8192) != hipSuccess ||
        hipMemcpy(d_tab, t.data(), sizeof(aacg_tables), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_in, in, 4096, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(d_out, 0, 8192) != hipSuccess) rc = AACG_ERR_NO_DEVICE;
    if (rc == AACG_OK) {
        hipLaunchKernelGGL(aacg_debug_transform_kernel, dim3(1), dim3(64), (AACG_TAB_F32_FLOATS + 1024) * sizeof(float), 0, d_tab, d_in, d_out, is_short);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d_out, 8192, hipMemcpyDeviceToHost) != hipSuccess) rc = AACG_ERR_NO_DEVICE;
    }
    if (d_tab) (void)hipFree(d_tab);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

/* Calibration for bench.py: a float4 copy with the run kernels' launch shape (one 1024-thread workgroup per CU, 16-byte
 * loads, 16-byte non-temporal stores) — what this box's memory system delivers for a launch of a given byte volume, timed
 * beside the run kernel in the same process.  Nothing on the decode path calls this. */
extern "C" __global__ __launch_bounds__(1024)
void aacg_calib_copy_kernel(dpf4* dst, const dpf4* src, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dp_store_nt(dst + i, src[i]);
}

extern "C" int aacg_calib_copy(void* d_dst, const void* d_src, size_t bytes, void* hip_stream)
{
    if (!d_dst || !d_src || (bytes & 15u)) return AACG_ERR_INVALID_ARG;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL(aacg_calib_copy_kernel, dim3((unsigned)cus), dim3(1024), 0, (hipStream_t)hip_stream, (dpf4*)d_dst, (const dpf4*)d_src, bytes / 16);
    return hipGetLastError() == hipSuccess ? AACG_OK : AACG_ERR_NO_DEVICE;
}

/* timing marks (include/aacgpu.h): events that only measure time */
extern "C" int aacg_timer_create(void** mark)
{
    if (!mark) return AACG_ERR_INVALID_ARG;
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableSystemFence) != hipSuccess) return AACG_ERR_NO_DEVICE;
    *mark = (void*)ev;
    return AACG_OK;
}
extern "C" int aacg_timer_record(void* mark, void* hip_stream)
{
    if (!mark) return AACG_ERR_INVALID_ARG;
    return hipEventRecord((hipEvent_t)mark, (hipStream_t)hip_stream) == hipSuccess ? AACG_OK : AACG_ERR_NO_DEVICE;
}
extern "C" int aacg_timer_elapsed_ms(void* first, void* second, float* ms)
{
    if (!first || !second || !ms) return AACG_ERR_INVALID_ARG;
    return hipEventElapsedTime(ms, (hipEvent_t)first, (hipEvent_t)second) == hipSuccess ? AACG_OK : AACG_ERR_NO_DEVICE;
}
extern "C" void aacg_timer_destroy(void* mark)
{
    if (mark) (void)hipEventDestroy((hipEvent_t)mark);
}
