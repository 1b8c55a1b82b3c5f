// Microbenchmark: issue cost of the cross-lane moves the register transposes are made of (profiling aid, not product code):
// v_mov_b32_dpp, v_cndmask_b32_dpp, v_permlane32_swap_b32, v_permlane16_swap_b32, against v_mov_b32 / v_pk_fma_f32.
// hipcc --offload-arch=gfx950 -O3 -o xlane_rate xlane_rate.hip && ./xlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
template <int MODE>
__global__ void k(float* out, unsigned long long* t)
{
    float a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; b[i] = 2.0f * threadIdx.x + i; }
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(a[i]) : "v"(b[i]));
            else if (MODE == 1) asm volatile("v_cndmask_b32_dpp %0, %1, %0, vcc row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            else if (MODE == 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            else if (MODE == 3) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            else if (MODE == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            else if (MODE == 6) asm volatile("v_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            else if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        }
    }
    unsigned long long t1 = wall_clock64();
    float r = 0; for (int i = 0; i < 8; i++) r += a[i] + b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { t[2 * (threadIdx.x >> 6)] = t0; t[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE> void run(const char* name, int threads)
{
    float* out; unsigned long long* t;
    hipMalloc(&out, 1 << 20); hipMalloc(&t, 4096);
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, out, t);
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(threads), 0, 0, out, t);
    hipDeviceSynchronize();
    unsigned long long h[32]; hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < threads / 64; w++) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
    double ns = (hi - lo) * 10.0;                                 // 100 MHz ticks
    double per = ns / ((double)ITER * 8);
    printf("%-22s threads %4d (waves/SIMD %.1f): %.3f ns per instruction per wave -> %.2f ns per instruction per SIMD\n",
           name, threads, threads / 256.0, per, per / (threads > 256 ? threads / 256.0 : 1.0));
    hipFree(out); hipFree(t);
}
int main()
{
    for (int th : {64, 256, 1024}) {
        run<4>("v_mov_b32", th); run<5>("v_fma_f32", th); run<0>("v_mov_b32_dpp ror8", th); run<7>("v_mov_b32_dpp quad", th); run<1>("v_cndmask_b32_dpp", th);
        run<2>("v_permlane32_swap", th); run<3>("v_permlane16_swap", th); run<6>("v_swap_b32", th);
    }
    return 0;
}
