// Which pairs of HIP streams run concurrently on MI355X / ROCm 7.2?  HIP multiplexes its streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default); two streams on one queue serialise whatever the program meant.  A process that has made many
// streams before (PyTorch creates 32 per priority at its first torch.cuda.Stream()) hands later streams whichever queue is least
// referenced.  This probe makes `crowd` plain streams first, then a pair of streams each way, and times two 128-workgroup kernels of
// ~40 us launched one on each: ~40 us = concurrent, ~80 us = serialised.   hipcc --offload-arch=gfx950 -O2 -o queue_map queue_map.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long ticks, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double pair_time(hipStream_t a, hipStream_t b)
{
    int* d; hipMalloc(&d, 4);
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipDeviceSynchronize();
        const double t0 = now();
        hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, a, 4000LL, d);   // 100 MHz clock: 40 us
        hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, b, 4000LL, d);
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        best = std::min(best, now() - t0);
    }
    hipFree(d);
    return best * 1e6;
}

int main(int argc, char** argv)
{
    const int crowd = argc > 1 ? atoi(argv[1]) : 40;
    std::vector<hipStream_t> others((size_t)crowd);
    int* d; hipMalloc(&d, 4);
    for (auto& s : others) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1LL, d); }
    hipDeviceSynchronize();
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::printf("crowd of %d streams first; priority range: least %d .. greatest %d\n", crowd, lo, hi);
    {
        hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        std::printf("two plain non-blocking streams:                 %.1f us\n", pair_time(a, b));
    }
    {
        hipStream_t a, b; hipStreamCreateWithPriority(&a, hipStreamNonBlocking, lo); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
        std::printf("one lowest- and one highest-priority stream:     %.1f us\n", pair_time(a, b));
    }
    {
        hipStream_t a, b; hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
        std::printf("two highest-priority streams:                    %.1f us\n", pair_time(a, b));
    }
    {
        hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
        const uint32_t words = (uint32_t)((p.multiProcessorCount + 31) / 32);
        std::vector<uint32_t> mask(words, 0xffffffffu);
        hipStream_t a, b;
        hipError_t ra = hipExtStreamCreateWithCUMask(&a, words, mask.data()), rb = hipExtStreamCreateWithCUMask(&b, words, mask.data());
        if (ra == hipSuccess && rb == hipSuccess) std::printf("two streams with an all-CUs mask (hipExtStreamCreateWithCUMask): %.1f us\n", pair_time(a, b));
        else std::printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(ra != hipSuccess ? ra : rb));
    }
    return 0;
}
