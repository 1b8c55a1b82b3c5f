// Microbenchmark (profiling aid, not product code): what does a launch of the run kernels' SHAPE cost when the kernel does nothing?
// The headline launch is 256 workgroups x 16 waves, ~152 KB of static LDS (one workgroup per CU), 120 VGPRs, three launches in
// flight on three streams.  tools/floor.sh found that skipping the dequantisation or the PCM stores shortens a launch behind a
// launch by 1.0 / 1.5 us and the overlapped route by nothing: something that does not depend on the kernel's work sets the period.
// Here every wave just lives for `life` microseconds (s_sleep on the constant 100 MHz clock), and launches go round three
// streams like aacg_decode_pipelined's: cost per launch against life, waves per workgroup, LDS per workgroup, streams.
//   hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS_BYTES, int THREADS>
__global__ __launch_bounds__(THREADS) void shape(unsigned* sink, int life_ticks, int jitter, int launch)
{
    // jitter: 1 = one workgroup in a hundred lives 4 us longer, 2 = one in ten 1 us longer, 3 = every workgroup 0..1 us longer (hashed)
    const unsigned h = (blockIdx.x * 2654435761u + (unsigned)launch * 40503u) >> 7;
    if ((jitter & 15) == 1 && h % 100u == 0) life_ticks += 400;
    if ((jitter & 15) == 2 && h % 10u == 0) life_ticks += 100;
    if ((jitter & 15) == 3) life_ticks += (int)(h % 100u);
    __shared__ unsigned lds[LDS_BYTES / 4];
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = threadIdx.x;                                  // the allocation is real
    asm volatile("v_mov_b32 v119, 0" ::: "v119");                    // 120 VGPRs like the run kernels
    // jitter bits 4..: what a run kernel does at its two ends.  16: copy 24 KB of tables from global memory into LDS and meet at a
    // barrier before anything else; 32: every wave ends with 8 KB of non-temporal 16-byte stores (its PCM); 64: ... and waits for them
    if (jitter & 16) {
        const uint4* tab = (const uint4*)(sink + 1024);
        for (int i = threadIdx.x; i < 1536; i += THREADS) ((uint4*)lds)[i + 64] = tab[i];
        __syncthreads();
    }
    if (!(jitter & 256)) while ((long long)(wall_clock64() - t0) < (long long)life_ticks) __builtin_amdgcn_s_sleep(1);
    if (jitter & 128) {           // 128: ONE 16-byte store per lane at the very end (does a wave's end wait for its last store's way out?)
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* out = (v4u*)(sink + 65536) + ((size_t)(launch & 7) * gridDim.x * (THREADS / 64) + (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 512 + (threadIdx.x & 63);
        const v4u v = {threadIdx.x, (unsigned)launch, 0u, 0u};
        if (jitter & 512) *out = v; else __builtin_nontemporal_store(v, out);
    }
    if (jitter & 256) {           // 256: the run kernels' rhythm — the four groups of waves end a quarter of the life apart, each with its 8 KB of stores
        const int group = (threadIdx.x / 64) / 4;
        const long long until = (long long)life_ticks * (group + 1) / 4;
        while ((long long)(wall_clock64() - t0) < until) __builtin_amdgcn_s_sleep(1);
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* out = (v4u*)(sink + 65536) + ((size_t)(launch & 7) * gridDim.x * (THREADS / 64) + (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 512 + (threadIdx.x & 63);
        const v4u v = {threadIdx.x, (unsigned)launch, 0u, 0u};
        // 1024: only the LAST group's stores are plain ones
        const bool plain = (jitter & 512) || ((jitter & 1024) && group == 3);
#pragma unroll
        for (int k = 0; k < 8; k++) { if (plain) out[64 * k] = v; else __builtin_nontemporal_store(v, out + 64 * k); }
        if (lds[(threadIdx.x + 1) % THREADS] == 0xffffffffu) sink[0] = 1;
        return;
    }
    if (jitter & 32) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* out = (v4u*)(sink + 65536) + ((size_t)(launch & 7) * gridDim.x * (THREADS / 64) + (size_t)blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 512 + (threadIdx.x & 63);
        const v4u v = {threadIdx.x, (unsigned)launch, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v, out + 64 * k);
        if (jitter & 64) __builtin_amdgcn_s_waitcnt(0);
    }
    if (lds[(threadIdx.x + 1) % THREADS] == 0xffffffffu) sink[0] = 1;
}

template <int LDS_BYTES, int THREADS>
static double run(int n_streams, hipStream_t* st, unsigned* sink, int grid, double life_us, int launches, int jitter = 0)
{
    const int ticks = (int)(life_us * 100.0);
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL((shape<LDS_BYTES, THREADS>), dim3(grid), dim3(THREADS), 0, st[i % n_streams], sink, ticks, jitter, i);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; i++)                               // no back-pressure: the queues hold the launches
        hipLaunchKernelGGL((shape<LDS_BYTES, THREADS>), dim3(grid), dim3(THREADS), 0, st[i % n_streams], sink, ticks, jitter, i);
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / launches;
}

int main()
{
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    unsigned* sink = nullptr;
    CK(hipMalloc(&sink, (size_t)320 << 20));              // tables at +4 KB, eight launches' worth of PCM at +256 KB
    CK(hipMemset(sink, 0, (size_t)320 << 20));
    const int N = 3000;
    const double lives[] = {0.0, 2.0, 4.0, 6.0, 8.0, 9.0, 10.0, 12.0};
    std::printf("cost of a launch that does nothing but live (us per launch; 256 workgroups unless said; %d launches, streams taken in turn)\n", N);
    std::printf("%-64s", "life of a wave, us:");
    for (double l : lives) std::printf(" %6.1f", l);
    std::printf("\n");
    auto row = [&](const char* name, auto fn) {
        std::printf("%-64s", name);
        for (double l : lives) std::printf(" %6.2f", fn(l));
        std::printf("\n");
        std::fflush(stdout);
    };
    row("16 waves, 152 KB LDS (the run kernels' shape), 3 streams", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N); });
    row("16 waves, 152 KB LDS, 2 streams", [&](double l) { return run<152 * 1024, 1024>(2, st, sink, 256, l, N); });
    row("16 waves, 152 KB LDS, 1 stream", [&](double l) { return run<152 * 1024, 1024>(1, st, sink, 256, l, N); });
    row("16 waves, 152 KB LDS, 4 streams", [&](double l) { return run<152 * 1024, 1024>(4, st, sink, 256, l, N); });
    row("16 waves, 76 KB LDS (two workgroups fit a CU), 3 streams", [&](double l) { return run<76 * 1024, 1024>(3, st, sink, 256, l, N); });
    row("16 waves, 4 KB LDS, 3 streams", [&](double l) { return run<4 * 1024, 1024>(3, st, sink, 256, l, N); });
    row("8 waves, 152 KB LDS, 3 streams", [&](double l) { return run<152 * 1024, 512>(3, st, sink, 256, l, N); });
    row("8 waves, 76 KB LDS, 512 workgroups, 3 streams", [&](double l) { return run<76 * 1024, 512>(3, st, sink, 512, l, N); });
    row("4 waves, 152 KB LDS, 3 streams", [&](double l) { return run<152 * 1024, 256>(3, st, sink, 256, l, N); });
    row("16 waves, 152 KB LDS, 128 workgroups, 3 streams", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 128, l, N); });
    row("16 waves, 152 KB LDS, 512 workgroups (two rounds), 3 streams", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 512, l, N); });
    row("the run kernels' shape, 3 streams, 24 KB of tables copied to LDS + a barrier first", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 16); });
    row("the run kernels' shape, 3 streams, every wave ends with 8 KB of nt stores", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 32); });
    row("the run kernels' shape, 3 streams, ... and waits for them", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 32 + 64); });
    row("the run kernels' shape, 3 streams, tables first and stores last", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 16 + 32); });
    row("the run kernels' shape, 3 streams, every wave ends with ONE 16-byte store per lane", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 128); });
    row("the run kernels' shape, 3 streams, groups of 4 waves end a quarter-life apart, 8 KB each", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 256); });
    row("... ONE 16-byte store per lane at the end, a plain (cached) store", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 128 + 512); });
    row("... groups a quarter-life apart, 8 KB each, plain stores", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 256 + 512); });
    row("... groups a quarter-life apart, 8 KB each, nt but the last group's plain", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 256 + 1024); });
    row("the run kernels' shape, 3 streams, one workgroup in 100 lives 4 us longer", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 1); });
    row("the run kernels' shape, 3 streams, one workgroup in 10 lives 1 us longer", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 2); });
    row("the run kernels' shape, 3 streams, every workgroup 0..1 us longer", [&](double l) { return run<152 * 1024, 1024>(3, st, sink, 256, l, N, 3); });
    row("the run kernels' shape, 4 streams, one workgroup in 100 lives 4 us longer", [&](double l) { return run<152 * 1024, 1024>(4, st, sink, 256, l, N, 1); });
    row("the run kernels' shape, 2 streams, one workgroup in 100 lives 4 us longer", [&](double l) { return run<152 * 1024, 1024>(2, st, sink, 256, l, N, 1); });
    return 0;
}
