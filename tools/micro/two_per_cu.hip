// Microbenchmark (profiling aid, not product code): what would TWO workgroups per CU be worth to a kernel like the run kernels?
// a launch costs a workgroup's life + 1.0-1.2 us (the dispatcher + the last stores' way out), and 1 us of the life is a
// ramp during which the SIMDs idle; a CU holds one 152 KB workgroup.  If the same work came as workgroups of 8 waves and <= 80 KB,
// two would share a CU and one could compute while the other ramps or drains.  Model: a wave idles for `ramp` us (sleep), issues
// `work` us of dependent-free VALU work (eight independent FMA chains; four waves per SIMD saturate it), and ends with 8 KB of
// non-temporal stores.  A: 256 workgroups x 16 waves, 152 KB LDS.  B: 512 x 8 waves, 76 KB.  Three streams in turn, no back-pressure.
//   hipcc --offload-arch=gfx950 -O2 -o two_per_cu two_per_cu.hip && ./two_per_cu
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS_BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void model(unsigned* sink, int ramp_ticks, int iters, int stores, int launch, int prio)
{
    __shared__ unsigned lds[LDS_BYTES / 4];
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v119, 0" ::: "v119");                    // 120 VGPRs like the run kernels
    while ((long long)(wall_clock64() - t0) < (long long)ramp_ticks) __builtin_amdgcn_s_sleep(1);
    const int wave = threadIdx.x >> 6;
    if (prio) { const int g = (wave >> 2) & 3; if (g == 0) __builtin_amdgcn_s_setprio(3); else if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
        a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
    }
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (stores) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* out = (v4u*)(sink + 65536) + ((size_t)(launch & 7) * 4096 + (size_t)blockIdx.x * (THREADS / 64) + wave) * 512 + (threadIdx.x & 63);
        const v4u v = {__float_as_uint(r), (unsigned)launch, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v, out + 64 * k);
    } else if (r == 123.456f) sink[0] = 1;
    if (lds[(threadIdx.x + 1) % THREADS] == 0xffffffffu) sink[1] = 1;
}

template <int LDS_BYTES, int THREADS>
static double run(hipStream_t* st, unsigned* sink, int grid, double ramp_us, int iters, int stores, int prio)
{
    const int N = 3000;
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL((model<LDS_BYTES, THREADS>), dim3(grid), dim3(THREADS), 0, st[i % 3], sink, (int)(ramp_us * 100), iters, stores, i, prio);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) hipLaunchKernelGGL((model<LDS_BYTES, THREADS>), dim3(grid), dim3(THREADS), 0, st[i % 3], sink, (int)(ramp_us * 100), iters, stores, i, prio);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
}

int main()
{
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st[3];
    for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    unsigned* sink = nullptr;
    CK(hipMalloc(&sink, (size_t)320 << 20));
    CK(hipMemset(sink, 0, (size_t)320 << 20));
    std::printf("a model of the run kernels' launch: per wave an idle ramp, VALU work that saturates a SIMD with four waves on it, 8 KB of stores (us per launch)\n");
    std::printf("%-78s %10s %10s\n", "", "A: 256 x 16 waves, 152 KB", "B: 512 x 8 waves, 76 KB");
    const int iters[] = {150, 75};         // 8 FMAs x iters per wave at 4 cycles each, four waves to a SIMD: about 8.7 us / 4.4 us of every SIMD per 16 waves
    for (int it : iters)
        for (int prio = 0; prio < 2; prio++)
            for (int stores = 0; stores < 2; stores++)
                for (double ramp : {0.0, 1.0, 2.0}) {
                    char name[160];
                    std::snprintf(name, sizeof name, "%d FMA x 8 per wave, ramp %.1f us, %s, %s", it, ramp, stores ? "8 KB of stores per wave" : "no stores", prio ? "priorities by group" : "flat priorities");
                    const double a = run<152 * 1024, 1024>(st, sink, 256, ramp, it, stores, prio);
                    const double b = run<76 * 1024, 512>(st, sink, 512, ramp, it, stores, prio);
                    std::printf("%-78s %10.2f %25.2f\n", name, a, b);
                    std::fflush(stdout);
                }
    return 0;
}
