// Microbenchmark (profiling aid, not product code): what does 32 MiB of FRESH page-locked host memory cost, and can it be had
// cheaper?  The plugin surface's default mode hands every flush memory of its own (what readChunk() returns is the caller's),
// and a synchronous drain never recycles it: the allocation is the largest item of a flush.
//   hipHostMalloc with each flag; mmap + MADV_HUGEPAGE + touch + hipHostRegister; two threads allocating at once; and what a
//   kernel launch on another thread waits while an allocation is in progress (the runtime's locks).
//   hipcc --offload-arch=gfx950 -O2 -o pinned_alloc pinned_alloc.hip -lpthread && ./pinned_alloc
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void nop(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
static const size_t B = 32u << 20;

static void line(const char* what, std::vector<double> v) { std::printf("%-86s", what); for (double x : v) std::printf(" %7.2f", x * 1e3); std::printf("  ms\n"); std::fflush(stdout); }

int main()
{
    (void)hipFree(nullptr);
    { FILE* f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = ""; if (f) { if (std::fgets(b, sizeof b, f)) std::printf("transparent_hugepage/enabled: %s", b); std::fclose(f); } }
    const unsigned flags[] = {hipHostMallocDefault, hipHostMallocPortable, hipHostMallocMapped, hipHostMallocNonCoherent, hipHostMallocCoherent, hipHostMallocNumaUser};
    const char* names[] = {"hipHostMalloc Default", "hipHostMalloc Portable", "hipHostMalloc Mapped", "hipHostMalloc NonCoherent", "hipHostMalloc Coherent", "hipHostMalloc NumaUser"};
    for (int k = 0; k < 6; k++) {
        std::vector<double> a, f;
        for (int r = 0; r < 5; r++) {
            void* p = nullptr; double t0 = now();
            if (hipHostMalloc(&p, B, flags[k]) != hipSuccess) { (void)hipGetLastError(); a.push_back(-1e-3); f.push_back(0); continue; }
            double t1 = now(); (void)hipHostFree(p); double t2 = now();
            a.push_back(t1 - t0); f.push_back(t2 - t1);
        }
        line((std::string(names[k]) + " 32 MiB, five times: allocate").c_str(), a); line("   ... free", f);
    }
    for (int huge = 0; huge < 2; huge++) {
        std::vector<double> m, t, g, u;
        for (int r = 0; r < 5; r++) {
            double t0 = now();
            char* p = (char*)mmap(nullptr, B + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            char* q = (char*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
            if (huge) madvise(q, B, MADV_HUGEPAGE);
            double t1 = now();
            for (size_t i = 0; i < B; i += 4096) q[i] = 1;           // fault every page in
            double t2 = now();
            hipError_t e = hipHostRegister(q, B, hipHostRegisterDefault);
            double t3 = now();
            if (e == hipSuccess) (void)hipHostUnregister(q); else (void)hipGetLastError();
            munmap(p, B + (2u << 20));
            double t4 = now();
            m.push_back(t1 - t0); t.push_back(t2 - t1); g.push_back(e == hipSuccess ? t3 - t2 : -1e-3); u.push_back(t4 - t3);
        }
        line(huge ? "mmap + MADV_HUGEPAGE: map" : "mmap (4 KiB pages): map", m); line("   ... touch every page", t); line("   ... hipHostRegister", g); line("   ... unregister + unmap", u);
    }
    for (int nt = 2; nt <= 8; nt *= 2) {   // the same with the pages faulted in by several threads at once, then a copy down into it
        std::vector<double> t, g, c, u;
        void* d = nullptr; (void)hipMalloc(&d, B); hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        int type = -1;
        for (int r = 0; r < 5; r++) {
            char* p = (char*)mmap(nullptr, B + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            char* q = (char*)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
            madvise(q, B, MADV_HUGEPAGE);
            double t1 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back([=] { const size_t a = B / nt * k, b = k == nt - 1 ? B : B / nt * (k + 1); for (size_t i = a; i < b; i += 4096) q[i] = 1; });
            for (auto& x : th) x.join();
            double t2 = now();
            hipError_t e = hipHostRegister(q, B, hipHostRegisterDefault);
            double t3 = now();
            hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, q) == hipSuccess) type = (int)at.type; else (void)hipGetLastError();
            (void)hipMemcpyAsync(q, d, B, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
            double t4 = now();
            if (e == hipSuccess) (void)hipHostUnregister(q); else (void)hipGetLastError();
            munmap(p, B + (2u << 20));
            double t5 = now();
            t.push_back(t2 - t1); g.push_back(e == hipSuccess ? t3 - t2 : -1e-3); c.push_back(t4 - t3); u.push_back(t5 - t4);
        }
        char name[128]; std::snprintf(name, sizeof name, "mmap + MADV_HUGEPAGE, pages touched by %d threads: touch", nt);
        line(name, t); line("   ... hipHostRegister", g); line("   ... a 32 MiB copy down into it (hipMemcpyAsync + synchronize)", c); line("   ... unregister + unmap", u);
        std::printf("   ... hipPointerGetAttributes type of the registered memory: %d (hipMemoryTypeHost = %d)\n", type, (int)hipMemoryTypeHost);
        (void)hipFree(d); (void)hipStreamDestroy(s);
    }
    {   // the copy down into hipHostMalloc memory, for comparison
        void *d = nullptr, *h = nullptr; (void)hipMalloc(&d, B); (void)hipHostMalloc(&h, B, hipHostMallocDefault); hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        std::vector<double> c;
        for (int r = 0; r < 5; r++) { double t0 = now(); (void)hipMemcpyAsync(h, d, B, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); c.push_back(now() - t0); }
        line("a 32 MiB copy down into hipHostMalloc memory", c);
        (void)hipFree(d); (void)hipHostFree(h); (void)hipStreamDestroy(s);
    }
    {   // two threads allocating at once
        std::vector<double> both;
        for (int r = 0; r < 5; r++) {
            double t0 = now();
            void *a = nullptr, *b = nullptr;
            std::thread x([&] { (void)hipHostMalloc(&a, B, hipHostMallocDefault); }), y([&] { (void)hipHostMalloc(&b, B, hipHostMallocDefault); });
            x.join(); y.join();
            both.push_back(now() - t0);
            (void)hipHostFree(a); (void)hipHostFree(b);
        }
        line("two threads, one hipHostMalloc of 32 MiB each, at once: until both have theirs", both);
    }
    {   // a launch + synchronize on this thread while another thread allocates
        hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr);
        (void)hipStreamSynchronize(s);
        std::vector<double> quiet, busy;
        for (int r = 0; r < 5; r++) {
            double t0 = now();
            for (int i = 0; i < 20; i++) { hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipStreamSynchronize(s); }
            quiet.push_back((now() - t0) / 20);
            std::atomic<bool> stop{false};
            std::thread x([&] { while (!stop) { void* a = nullptr; if (hipHostMalloc(&a, B, hipHostMallocDefault) == hipSuccess) (void)hipHostFree(a); } });
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            double worst = 0;
            for (int i = 0; i < 20; i++) { double a0 = now(); hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s, (int*)nullptr); (void)hipStreamSynchronize(s); worst = std::max(worst, now() - a0); }
            stop = true; x.join();
            busy.push_back(worst);
        }
        line("launch + synchronize of an empty kernel, nothing else going on (mean of 20)", quiet);
        line("   ... the slowest of 20 while another thread allocates and frees 32 MiB in a loop", busy);
    }
    return 0;
}
