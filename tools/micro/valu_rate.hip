// Microbenchmark: issue cost of packed / scalar fp32 FMA chains on one CU (profiling aid, not product code).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 20000
template <int MODE, int ILP>
__global__ void k(float* out, unsigned long long* t)
{
    v2f a[ILP]; float s[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = (v2f){(float)threadIdx.x + i, 1.0f}; s[i] = threadIdx.x + i; }
    const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], m, c);       // v_pk_fma_f32
            else if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[0]), "v"(c[0]));   // not SLP-vectorised
            else if (MODE == 2) a[i] = a[i] * m;                                  // v_pk_mul_f32
            else a[i] = a[i] + c;                                                 // v_pk_add_f32
        }
    }
    unsigned long long t1 = wall_clock64();
    float r = 0; for (int i = 0; i < ILP; i++) r += a[i][0] + a[i][1] + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = t0; t[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE, int ILP> void run(const char* name, int threads)
{
    float* out; unsigned long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(1), dim3(threads), 0, 0, out, t);
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(1), dim3(threads), 0, 0, out, t);
    hipDeviceSynchronize();
    unsigned long long h[32]; hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < threads / 64; w++) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    double ns = (hi - lo) * 10.0;                                 // 100 MHz ticks, first start to last end over the waves
    double per = ns / ((double)ITER * ILP);
    printf("%-14s ilp %d threads %4d (waves/SIMD %.1f): %.3f ns per instruction per wave  -> %.2f ns per instr per SIMD\n",
           name, ILP, threads, threads / 256.0, per, per / (threads > 256 ? threads / 256.0 : 1.0));
    hipFree(out); hipFree(t);
}
int main()
{
    for (int th : {64, 256, 1024}) {
        run<0, 1>("pk_fma dep", th); run<0, 8>("pk_fma ilp8", th);
        run<1, 1>("fma dep", th);    run<1, 8>("fma ilp8", th);
        run<2, 8>("pk_mul ilp8", th); run<3, 8>("pk_add ilp8", th);
    }
    return 0;
}
