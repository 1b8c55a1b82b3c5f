// Microbenchmark (profiling aid, not product code): what HBM takes per second as stores alone, loads alone, and half and
// half — for the run kernel's launch shape (one 1024-thread workgroup per CU, 16-byte accesses, non-temporal stores) and
// its volumes (35.6 MB of PCM out, 22 MB of spectra in per 4096-frame batch).  Buffers rotate past the 256 MB MALL.
//   hipcc --offload-arch=gfx950 -O3 -o write_rate write_rate.hip && ./write_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void wr(v4f* dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const v4f v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, dst + i);
}
__global__ __launch_bounds__(1024) void rd(const v4f* src, size_t n4, float* sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += src[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
__global__ __launch_bounds__(1024) void cp(v4f* dst, const v4f* src, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(src[i], dst + i);
}
// Config 5's store pattern: PCM [frame][1024][7] float, an ELEMENT per workgroup (3 channel pairs + 1 single
// channel per frame; a wave = one frame of the element, 16 frames per workgroup), a store instruction = 64 consecutive sample-frames of
// the element's channels: 8 (or 4) bytes of every 28.  Blocks dealt out as the planner deals runs (consecutive runs 8 blocks apart: the
// four elements of a stream on one XCD, co-scheduled).  Nothing but the stores: what the CU write path + L2 + HBM take for this layout.
//   mode 0: element-major (the run kernels' epilogue);  mode 1: the same bytes, every wave a contiguous 28-byte-per-sample-frame stretch
//   (what a frame-major workgroup could store if its elements' samples met in LDS first)
//   mode 2 (round 6): TWO elements per workgroup — CPE0 + CPE1: a 16-byte piece of every 28, CPE2 + LFE: a
//   12-byte piece — half the workgroups (2 per frame), half the chains; what the write path takes for that pattern alone
__global__ __launch_bounds__(1024) void wr7b(float* pcm, int n_streams, int nt)
{
    const int R = 2 * n_streams, b = blockIdx.x, x = b & 7, per = R >> 3;
    const int i = x * per + (b >> 3), s = i >> 1, half = i & 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* base = pcm + ((size_t)s * 16 + w) * 7168 + 4 * half;
    for (int j = 0; j < 16; j++) {
        const int n = lane + 64 * j;
        float* p = base + (size_t)n * 7;
        if (!half) { typedef float f4u __attribute__((ext_vector_type(4), aligned(4))); f4u v = {1.0f, (float)n, 2.0f, 3.0f}; if (nt) __builtin_nontemporal_store(v, (f4u*)p); else *(f4u*)p = v; }
        else { typedef float f2u __attribute__((ext_vector_type(2), aligned(4))); f2u v = {1.0f, (float)n}; if (nt) { __builtin_nontemporal_store(v, (f2u*)p); __builtin_nontemporal_store(3.0f, p + 2); } else { *(f2u*)p = v; p[2] = 3.0f; } }
    }
}
__global__ __launch_bounds__(1024) void wr7(float* pcm, int n_streams, int mode, int nt)
{
    const int R = 4 * n_streams, b = blockIdx.x, x = b & 7, per = R >> 3;
    const int i = x * per + (b >> 3), s = i >> 2, e = i & 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* frame = pcm + ((size_t)s * 16 + w) * 7168;
    if (mode == 0) {
        float* base = frame + 2 * e;
        for (int j = 0; j < 16; j++) {
            const int n = lane + 64 * j;
            float* p = base + (size_t)n * 7;
            if (e < 3) { typedef float f2u __attribute__((ext_vector_type(2), aligned(4))); f2u v = {1.0f, (float)n}; if (nt) __builtin_nontemporal_store(v, (f2u*)p); else *(f2u*)p = v; }
            else { if (nt) __builtin_nontemporal_store((float)n, p); else *p = (float)n; }
        }
    } else {
        /* element e's quarter of the frame's 7168 floats, contiguous: 28 float4 stores per wave-quarter */
        v4f* q = (v4f*)(frame + 1792 * e);
        const v4f v = {1.0f, 2.0f, 3.0f, (float)lane};
        for (int j = 0; j < 7; j++) { if (nt) __builtin_nontemporal_store(v, q + lane + 64 * j); else q[lane + 64 * j] = v; }
    }
}

int main()
{
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int SETS = 16;
    const size_t big = 64u << 20;                          // bytes per buffer set member
    char* buf; hipMalloc(&buf, big * SETS * 2);            // 2 GiB: also 8 sets of 128 MB + room for the config-5 pattern hipMemset(buf, 1, big * SETS * 2);
    float* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](int kind, size_t bytes) {
        const size_t n4 = bytes / 16;
        const int reps = 400;
        for (int pass = 0; pass < 2; pass++) {
            if (pass) hipEventRecord(e0);
            for (int i = 0; i < reps; i++) {
                v4f* a = (v4f*)(buf + (size_t)(i % SETS) * big);
                v4f* b = (v4f*)(buf + big * SETS + (size_t)(i % SETS) * big);
                if (kind == 0) hipLaunchKernelGGL(wr, dim3(cus), dim3(1024), 0, 0, a, n4);
                else if (kind == 1) hipLaunchKernelGGL(rd, dim3(cus), dim3(1024), 0, 0, a, n4, sink);
                else hipLaunchKernelGGL(cp, dim3(cus), dim3(1024), 0, 0, b, a, n4);
            }
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3 / reps;
    };
    const size_t pcm = 4096u * 2 * 1024 * 4 + 2 * 1024 * 1024, in = 4096u * 2 * 1024 * 2 + 5 * 1024 * 1024;   // ~35.6 MB, ~22 MB
    const double tw = timeit(0, pcm), tr = timeit(1, in), tc = timeit(2, (pcm + in) / 2);
    printf("stores alone %.1f MB: %.2f us = %.2f TB/s | loads alone %.1f MB: %.2f us = %.2f TB/s | copy of %.1f MB (read + written %.1f MB): %.2f us = %.2f TB/s\n",
           pcm / 1e6, tw, pcm / tw / 1e6, in / 1e6, tr, in / tr / 1e6, (pcm + in) / 2e6, (pcm + in) / 1e6, tc, (pcm + in) / tc / 1e6);
    const double tw1 = timeit(0, 1u << 30 > big ? big : (1u << 30)), tr1 = timeit(1, big);
    printf("64 MiB: stores alone %.2f TB/s, loads alone %.2f TB/s\n", big / tw1 / 1e6, big / tr1 / 1e6);
    /* config 5: 256 streams x 16 frames x 7 channels of float PCM = 117.4 MB per batch; 8 buffer sets of 128 MB rotate past the MALL */
    {
        const int S = 256, reps = 200;
        const size_t set = 128u << 20, bytes = (size_t)S * 16 * 7168 * 4;
        auto time7 = [&](int mode, int nt) {
            for (int pass = 0; pass < 2; pass++) {
                if (pass) hipEventRecord(e0);
                for (int i = 0; i < reps; i++) hipLaunchKernelGGL(wr7, dim3(4 * S), dim3(1024), 0, 0, (float*)(buf + (size_t)(i % 8) * set), S, mode, nt);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms * 1e3 / reps;
        };
        auto time7b = [&](int nt) {
            for (int pass = 0; pass < 2; pass++) {
                if (pass) hipEventRecord(e0);
                for (int i = 0; i < reps; i++) hipLaunchKernelGGL(wr7b, dim3(2 * S), dim3(1024), 0, 0, (float*)(buf + (size_t)(i % 8) * set), S, nt);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms * 1e3 / reps;
        };
        const double a = time7(0, 0), an = time7(0, 1), c = time7(1, 0), cn = time7(1, 1), p2 = time7b(0), p2n = time7b(1);
        printf("config-5 PCM stores alone, two elements per workgroup (512 workgroups: 16-byte and 12-byte pieces of every 28): %.2f us plain, %.2f us nt (%.2f TB/s)\n", p2, p2n, bytes / p2n / 1e6);
        printf("config-5 PCM stores alone (%.1f MB, 1024 workgroups): element-major 8/4 bytes of every 28: %.2f us plain (%.2f TB/s), %.2f us nt | "
               "the same bytes as contiguous 16-byte stores: %.2f us plain, %.2f us nt (%.2f TB/s)\n", bytes / 1e6, a, bytes / a / 1e6, an, c, cn, bytes / cn / 1e6);
        /* config 5's algorithmic bytes (197.7 MB) as a copy, half read and half written: 4 sets of 128 MB on either side */
        const size_t n4 = 98850000 / 16;
        for (int pass = 0; pass < 2; pass++) {
            if (pass) hipEventRecord(e0);
            for (int i = 0; i < reps; i++)
                hipLaunchKernelGGL(cp, dim3(cus), dim3(1024), 0, 0, (v4f*)(buf + (1u << 30) + (size_t)(i % 4) * set), (const v4f*)(buf + (size_t)(i % 4) * set), n4);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float cms; hipEventElapsedTime(&cms, e0, e1);
        const double tcopy = cms * 1e3 / reps;
        printf("copy of config 5's byte volume (197.7 MB read + written): %.2f us = %.2f TB/s\n", tcopy, 197.7e6 / tcopy / 1e6);
    }
    return 0;
}
