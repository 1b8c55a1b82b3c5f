// Microbenchmark (profiling aid, not product code): does a CU get through an IMDCT-like instruction mix faster with 8 waves
// per SIMD of half the work each than with 4 waves per SIMD?  One "phase" = a radix-8-like butterfly on 8 packed values
// (24 v_pk_add/fma with the dependency depth of three butterfly levels + 7 twiddle multiplies) and an exchange through
// LDS (2 x ds_write_b128, wave sync, 2 x ds_read_b128) whose result the next phase depends on — what the run kernel's
// IMDCT looks like to the issue logic.  A workgroup is 16 waves; its LDS request decides whether one or two fit on a CU.
//   hipcc --offload-arch=gfx950 -O3 -o occupancy occupancy.hip && ./occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
extern __shared__ float lds[];

__global__ __launch_bounds__(1024) void k(float* out, int phases, int slot_floats)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* slot = lds + wave * slot_floats;
    v2f a[8];
    for (int i = 0; i < 8; i++) a[i] = (v2f){(float)(threadIdx.x + i) * 1e-3f, 1.0f + i * 1e-3f};
    const v2f tw = {0.70710678f, 0.70710678f};
    for (int p = 0; p < phases; p++) {
        // three butterfly levels
        v2f b[8], c[8];
#pragma unroll
        for (int i = 0; i < 4; i++) { b[i] = a[i] + a[i + 4]; b[i + 4] = a[i] - a[i + 4]; }
#pragma unroll
        for (int i = 0; i < 8; i += 4) { c[i] = b[i] + b[i + 2]; c[i + 1] = b[i + 1] + b[i + 3]; c[i + 2] = b[i] - b[i + 2]; c[i + 3] = b[i + 1] - b[i + 3]; }
#pragma unroll
        for (int i = 0; i < 8; i += 2) { a[i] = c[i] + c[i + 1]; a[i + 1] = c[i] - c[i + 1]; }
        // twiddles
#pragma unroll
        for (int i = 1; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], tw, a[i - 1] * (v2f){1e-3f, -1e-3f});
        // exchange through LDS (XOR-swizzled like the kernel's transposes)
        v4f w0 = {a[0][0], a[0][1], a[1][0], a[1][1]}, w1 = {a[2][0], a[2][1], a[3][0], a[3][1]};
        *(v4f*)(slot + 4 * (lane ^ (p & 7))) = w0;
        *(v4f*)(slot + 256 + 4 * (lane ^ ((p + 3) & 7))) = w1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const v4f r0 = *(const v4f*)(slot + 4 * ((lane + 8) & 63)), r1 = *(const v4f*)(slot + 256 + 4 * ((lane + 24) & 63));
        a[4] += (v2f){r0[0], r0[1]}; a[5] += (v2f){r0[2], r0[3]}; a[6] += (v2f){r1[0], r1[1]}; a[7] += (v2f){r1[2], r1[3]};
        __builtin_amdgcn_wave_barrier();
    }
    float r = 0;
    for (int i = 0; i < 8; i++) r += a[i][0] + a[i][1];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double run(int blocks, int lds_bytes, int phases)
{
    float* out; hipMalloc(&out, (size_t)blocks * 1024 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), lds_bytes, 0, out, phases, 512);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), lds_bytes, 0, out, phases, 512);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e3 / reps;
}

int main()
{
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int P = 2400;                                    // phases per wave at 4 waves per SIMD (long enough to swamp the launch)
    // same work per CU: 16 waves x P phases, as one workgroup per CU (4 waves per SIMD) or as two of half the phases (8 per SIMD)
    const double t4 = run(cus, 100 * 1024, P);             // 100 KB of LDS: one workgroup per CU
    const double t8 = run(2 * cus, 64 * 1024, P / 2);      // 64 KB: two per CU
    const double t8s = run(2 * cus, 100 * 1024, P / 2);    // control: the same split, but one workgroup per CU at a time (two rounds)
    printf("4 waves/SIMD, %d phases each: %.1f us | 8 waves/SIMD, %d phases each: %.1f us (%.2fx) | split but not co-resident: %.1f us\n",
           P, t4, P / 2, t8, t4 / t8, t8s);
    return 0;
}
