// What does it cost to use a caller's pageable buffers in place (hipHostRegister / hipHostUnregister around the
// transfer) against staging them through pinned memory with memcpy?  The question behind aacg_submit's pageable path.
//   hipcc -O2 tools/micro/hostreg.hip -o tools/micro/hostreg && tools/micro/hostreg
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t in_b = 18743296, out_b = 33554432;         // config 2: int16 spectra + band words in, f32 PCM out
    void *d_in, *d_out, *pin_in, *pin_out;
    hipMalloc(&d_in, in_b); hipMalloc(&d_out, out_b);
    hipHostMalloc(&pin_in, in_b); hipHostMalloc(&pin_out, out_b);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; rep++) {
        char* in = (char*)malloc(in_b); char* out = (char*)malloc(out_b);
        memset(in, 1, in_b);                                 // the caller filled its input; its output array is fresh
        double t0 = now();
        memcpy(pin_in, in, in_b);
        hipMemcpyAsync(d_in, pin_in, in_b, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(pin_out, d_out, out_b, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        memcpy(out, pin_out, out_b);
        double t1 = now();
        free(out); out = (char*)malloc(out_b);
        double t2 = now();
        hipHostRegister(in, in_b, hipHostRegisterDefault);
        hipHostRegister(out, out_b, hipHostRegisterDefault);
        double t3 = now();
        hipMemcpyAsync(d_in, in, in_b, hipMemcpyHostToDevice, s);
        hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        double t4 = now();
        hipHostUnregister(in); hipHostUnregister(out);
        double t5 = now();
        free(out); out = (char*)malloc(out_b);
        double t6 = now();
        hipMemcpy(d_in, in, in_b, hipMemcpyHostToDevice);   // what the runtime does with pageable memory by itself
        hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost);
        double t7 = now();
        printf("staged through pinned memory %.2f ms | registered in place: register %.2f + transfers %.2f + unregister %.2f = %.2f ms | plain hipMemcpy of pageable memory %.2f ms\n",
               (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t2) * 1e3, (t7 - t6) * 1e3);
        free(in); free(out);
    }
    return 0;
}
