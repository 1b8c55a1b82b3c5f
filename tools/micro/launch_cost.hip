// Host cost of the calls aacg_decode_pipelined makes per launch (MI355X, ROCm 7.2): a kernel launch on two HIP streams taken in
// turn, alone and with the cross-stream ordering around it — hipEventRecord + hipStreamWaitEvent per launch, per q-th launch,
// the event bound to the dispatch (hipExtLaunchKernel stopEvent), stream memory operations.  Host wall time per launch while the
// queues are never full (the kernel is ~1 us).   hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 123456.f) p[1] = 1.f; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    float* d; hipMalloc(&d, 1024);
    hipStream_t s[2]; for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[8]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    hipEvent_t evd[8]; for (auto& e : evd) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 20000;
    void* args[1] = {&d};
    auto run = [&](const char* name, int mode, int q) {
        hipDeviceSynchronize();
        const double t0 = now();
        for (int n = 0; n < N; n++) {
            hipStream_t st = s[n & 1];
            const bool sync_point = ((n >> 1) % q) == 0;
            if (mode >= 1 && mode <= 3 && n >= 3 && sync_point) hipStreamWaitEvent(st, (mode == 3 ? evd : ev)[(n - 3) & 7], 0);
            const bool rec = (((n + 3) >> 1) % q) == 0;
            if (mode == 2 && rec) hipExtLaunchKernel((const void*)tiny, dim3(256), dim3(1024), args, 0, st, nullptr, ev[n & 7], 0);
            else hipLaunchKernel((const void*)tiny, dim3(256), dim3(1024), args, 0, st);
            if ((mode == 1 || mode == 3) && rec) hipEventRecord((mode == 3 ? evd : ev)[n & 7], st);
            if ((n & 1023) == 1023) hipDeviceSynchronize();   // keep the queues from filling up: host cost only
        }
        const double t1 = now();
        hipDeviceSynchronize();
        std::printf("%-70s q=%d  %.2f us per launch (host)\n", name, q, (t1 - t0) / N * 1e6);
    };
    run("launch only, two streams in turn", 0, 1);
    for (int q : {1, 2, 4, 8}) run("launch + hipEventRecord + hipStreamWaitEvent(n-3) [no-fence events]", 1, q);
    for (int q : {1, 2, 4}) run("launch with stopEvent (hipExtLaunchKernel) + hipStreamWaitEvent(n-3)", 2, q);
    run("launch + hipEventRecord + hipStreamWaitEvent(n-3) [default DisableTiming events]", 3, 1);
    {   // one stream, for reference
        hipDeviceSynchronize();
        const double t0 = now();
        for (int n = 0; n < N; n++) { hipLaunchKernel((const void*)tiny, dim3(256), dim3(1024), args, 0, s[0]); if ((n & 1023) == 1023) hipDeviceSynchronize(); }
        std::printf("%-70s      %.2f us per launch (host)\n", "launch only, one stream", (now() - t0) / N * 1e6);
    }
    return 0;
}
