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
int main()
{
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int SETS = 16;
    const size_t big = 64u << 20;                          // bytes per buffer set member
    char* buf; hipMalloc(&buf, big * SETS * 2); hipMemset(buf, 1, big * SETS * 2);
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
    return 0;
}
