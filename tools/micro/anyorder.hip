// Do two launches on ONE HIP stream overlap when the second is launched with hipExtAnyOrderLaunch?  And on TWO streams?
// Kernel A spins (bounded) until kernel B has raised a flag; it reports whether it saw the flag.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder tools/micro/anyorder.hip && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void waiter(int* flag, int* saw, long long budget)
{
    const long long t0 = wall_clock64();
    int v = 0;
    while (!(v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < budget) __builtin_amdgcn_s_sleep(10);
    *saw = v;
}
__global__ void raiser(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

static int run(bool any_order, bool two_streams)
{
    int *flag, *saw, h = -1;
    hipStream_t s1, s2;
    (void)hipMalloc(&flag, 4); (void)hipMalloc(&saw, 4); (void)hipMemset(flag, 0, 4); (void)hipMemset(saw, 0xff, 4);
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    (void)hipDeviceSynchronize();
    const long long budget = 100000000LL / 10;                 // wall_clock64 ticks at 100 MHz: 100 ms
    hipLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, s1, flag, saw, budget);
    if (any_order) hipExtLaunchKernelGGL(raiser, dim3(1), dim3(64), 0, two_streams ? s2 : s1, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
    else hipLaunchKernelGGL(raiser, dim3(1), dim3(64), 0, two_streams ? s2 : s1, flag);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, saw, 4, hipMemcpyDeviceToHost);
    (void)hipFree(flag); (void)hipFree(saw); (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
    return h;
}

int main()
{
    printf("one stream, ordered launch   : waiter saw the flag = %d (expect 0)\n", run(false, false));
    printf("one stream, any-order launch : waiter saw the flag = %d\n", run(true, false));
    printf("two streams                  : waiter saw the flag = %d (expect 1)\n", run(false, true));
    return 0;
}
