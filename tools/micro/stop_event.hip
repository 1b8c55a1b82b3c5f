// Does an event bound to a dispatch with hipExtLaunchKernel's stopEvent order another stream like a recorded event does, and does it
// carry the dispatch's end as its time stamp?  (aacg_decode_pipelined binds its ordering events and bench.py's timing marks that
// way: no marker packet in the queue between two launches.)   hipcc --offload-arch=gfx950 -O2 -o stop_event stop_event.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(long long ticks, long long* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wall_clock64(); }
__global__ void stamp(long long* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall_clock64(); }

int main()
{
    long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t a, b; hipStreamCreateWithPriority(&a, hipStreamNonBlocking, -1); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1);
    int bad = 0;
    for (int flags : {(int)hipEventDisableTiming | (int)hipEventDisableSystemFence, (int)hipEventDisableSystemFence}) {
        hipEvent_t e, e0; hipEventCreateWithFlags(&e, flags); hipEventCreateWithFlags(&e0, flags);
        for (int rep = 0; rep < 20; rep++) {
            long long ticks = 20000;                                   // 200 us
            void* args[2] = {&ticks, &d};
            hipExtLaunchKernel((const void*)spin, dim3(64), dim3(64), args, 0, a, nullptr, e, 0);
            hipStreamWaitEvent(b, e, 0);
            hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, d);
            hipDeviceSynchronize();
            long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            if (h[1] < h[0]) bad++;
        }
        std::printf("flags 0x%x: waiter started before the dispatch ended in %d of 20 trials\n", flags, bad);
        if (!(flags & hipEventDisableTiming)) {
            long long t1 = 10000, t2 = 30000; void* a1[2] = {&t1, &d}; void* a2[2] = {&t2, &d};
            hipExtLaunchKernel((const void*)spin, dim3(64), dim3(64), a1, 0, a, nullptr, e0, 0);
            hipExtLaunchKernel((const void*)spin, dim3(64), dim3(64), a2, 0, a, nullptr, e, 0);
            hipDeviceSynchronize();
            float ms = 0; hipError_t rc = hipEventElapsedTime(&ms, e0, e);
            std::printf("elapsed between the stop events of a 100 us and a following 300 us dispatch: %.1f us (%s)\n", ms * 1e3, hipGetErrorString(rc));
        }
    }
    return bad != 0;
}
