/*
 * devport_emu.h — CPU stand-in for aac.js_amd/csrc/devport.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Runs the very same kernel source (aacg_kernels.h) with one OS thread per lane:
 * 64 threads form a wavefront, (W+1)*64 a workgroup; dp_wave_sync / dp_shfl / dp_block_sync
 * are pthread barriers.  Lanes only interact at those points in the real kernel too, so an
 * indexing, twiddle, windowing or hand-off mistake shows up here — in a container without a
 * GPU (LDS is poisoned with NaN bits before every workgroup, so a read of a slot nobody wrote
 * is visible too).  What it cannot show: compiler/hardware issues (the -m gpu tests do).
 * Never compiled into the product library; selected by -DAACG_EMU_BUILD in tests/emu/Makefile.
 */
#ifndef AACG_DEVPORT_EMU_H
#define AACG_DEVPORT_EMU_H

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <string.h>
#include <stddef.h>

#define DP_DEVICE static inline
#define DP_KERNEL(a, b)

struct alignas(8)  dpf2 { float x, y; };
struct alignas(16) dpd2 { double x, y; };
struct alignas(16) dpf4 { float x, y, z, w; };
struct alignas(16) dpi4 { int x, y, z, w; };
struct alignas(8)  dpu2 { uint32_t x, y; };
typedef float dpv2 __attribute__((vector_size(8)));

struct emu_wave {
    pthread_barrier_t bar;
    float shfl[64][16];
    double shfl_d[64][16];
};
struct emu_block {
    pthread_barrier_t bar;
    unsigned char* lds;
    size_t lds_bytes;
    int block_id;
};
struct emu_lane_ctx {
    int lane, wave;
    emu_wave* w;
    emu_block* b;
};
extern thread_local emu_lane_ctx g_emu;

DP_DEVICE int dp_tid()   { return g_emu.wave * 64 + g_emu.lane; }
DP_DEVICE int dp_lane()  { return g_emu.lane; }
DP_DEVICE int dp_wave()  { return g_emu.wave; }
DP_DEVICE int dp_block() { return g_emu.b->block_id; }
DP_DEVICE int dp_uniform(int v) { return v; }
DP_DEVICE void dp_wave_sync()  { pthread_barrier_wait(&g_emu.w->bar); }
DP_DEVICE void dp_block_sync() { pthread_barrier_wait(&g_emu.b->bar); }
DP_DEVICE void dp_block_sync_lds() { pthread_barrier_wait(&g_emu.b->bar); }
DP_DEVICE void dp_flag_set(int* flag, int v) { __atomic_store_n(flag, v, __ATOMIC_RELEASE); }
DP_DEVICE void dp_flag_wait(int* flag, int v) { while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != v) sched_yield(); }
DP_DEVICE void dp_setprio(int) {}

template <int N>
DP_DEVICE void dp_shfl(float (&v)[N], int src)
{
    static_assert(N <= 16, "shuffle payload");
    for (int i = 0; i < N; i++) g_emu.w->shfl[g_emu.lane][i] = v[i];
    pthread_barrier_wait(&g_emu.w->bar);
    for (int i = 0; i < N; i++) v[i] = g_emu.w->shfl[src][i];
    pthread_barrier_wait(&g_emu.w->bar);
}

template <int N> DP_DEVICE void dp_mirror8_valu(const float (&in)[N], float (&out)[N]) { for (int i = 0; i < N; i++) out[i] = in[i]; dp_shfl(out, g_emu.lane ^ 7); }
template <int N> DP_DEVICE void dp_mirror16_valu(const float (&in)[N], float (&out)[N]) { for (int i = 0; i < N; i++) out[i] = in[i]; dp_shfl(out, g_emu.lane ^ 15); }

/* out[2k + c] = s_k * (mirror lane's src[2k + c]) * w[k], s = (-, -, -, +)  (devport.h) */
template <int MIRROR>
DP_DEVICE void dp_window_mirror(const float (&src)[8], const float (&w)[4], float (&out)[8], bool)
{
    float m[8];
    for (int i = 0; i < 8; i++) m[i] = src[i];
    dp_shfl(m, g_emu.lane ^ (MIRROR - 1));
    for (int i = 0; i < 8; i++) out[i] = (i < 6 ? -m[i] : m[i]) * w[i >> 1];
}

/* out[k] <- the value v of lane k of this lane's row of sixteen (devport.h) */
DP_DEVICE void dp_row_gather12(float v, float (&out)[12])
{
    g_emu.w->shfl[g_emu.lane][0] = v;
    pthread_barrier_wait(&g_emu.w->bar);
    for (int k = 0; k < 12; k++) out[k] = g_emu.w->shfl[(g_emu.lane & ~15) | k][0];
    pthread_barrier_wait(&g_emu.w->bar);
}

template <int N>
DP_DEVICE void dp_shfl(double (&v)[N], int src)
{
    static_assert(N <= 16, "shuffle payload");
    for (int i = 0; i < N; i++) g_emu.w->shfl_d[g_emu.lane][i] = v[i];
    pthread_barrier_wait(&g_emu.w->bar);
    for (int i = 0; i < N; i++) v[i] = g_emu.w->shfl_d[src][i];
    pthread_barrier_wait(&g_emu.w->bar);
}

/* one complex value per dpv2 (devport.h) */
DP_DEVICE dpv2 dp_cswap(dpv2 a) { dpv2 r; r[0] = a[1]; r[1] = a[0]; return r; }
DP_DEVICE dpv2 dp_cadd_i(dpv2 a, dpv2 b) { dpv2 r; r[0] = a[0] - b[1]; r[1] = a[1] + b[0]; return r; }
DP_DEVICE dpv2 dp_csub_i(dpv2 a, dpv2 b) { dpv2 r; r[0] = a[0] + b[1]; r[1] = a[1] - b[0]; return r; }
DP_DEVICE dpv2 dp_cmul(dpv2 a, dpv2 w)
{
    dpv2 r;
    r[0] = fmaf(-a[1], w[1], a[0] * w[0]);
    r[1] = fmaf(a[0], w[1], a[1] * w[0]);
    return r;
}
DP_DEVICE void dp_flag_wait_ge(int* flag, int v) { while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < v) sched_yield(); }
/* the run-to-run rendezvous through global memory (devport.h): plain host memory here */
typedef unsigned long long dp_u64;
DP_DEVICE dp_u64 dp_g_load_u64(const dp_u64* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
DP_DEVICE bool dp_g_cas_u64(dp_u64* p, dp_u64 expected, dp_u64 desired)
{
    return __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
}
DP_DEVICE void dp_g_store_f2(float* p, float a, float b) { p[0] = a; p[1] = b; }
DP_DEVICE dpf2 dp_g_load_f2(const float* p) { dpf2 v; v.x = p[0]; v.y = p[1]; return v; }
DP_DEVICE float dp_g_load_f1(const float* p) { return *p; }
DP_DEVICE void dp_g_store_u64(dp_u64* p, dp_u64 v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
DP_DEVICE unsigned dp_g_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
DP_DEVICE void dp_g_store_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
DP_DEVICE unsigned dp_cu_id() { return 0; }
/* scalar loads in a spelled-out batch (devport.h): here plain reads */
struct dp_su4 { unsigned v[4]; unsigned operator[](int i) const { return v[i]; } unsigned& operator[](int i) { return v[i]; } };
struct dp_su8 { unsigned v[8]; unsigned operator[](int i) const { return v[i]; } unsigned& operator[](int i) { return v[i]; } };
DP_DEVICE unsigned dp_sload1(const void* p, int byte_off) { unsigned r; memcpy(&r, (const char*)p + byte_off, 4); return r; }
DP_DEVICE dp_su4 dp_sload4(const void* p) { dp_su4 r; memcpy(r.v, p, 16); return r; }
DP_DEVICE dp_su8 dp_sload8(const void* p) { dp_su8 r; memcpy(r.v, p, 32); return r; }
DP_DEVICE void dp_swait(dp_su8&, dp_su4&, unsigned&, unsigned&, unsigned&) {}
DP_DEVICE dp_su8 dp_sload8(const void* p, const void*) { return dp_sload8(p); }
DP_DEVICE void dp_vm_drain() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
/* lane 0's value in every lane */
DP_DEVICE dp_u64 dp_first_u64(dp_u64 v)
{
    if (g_emu.lane == 0) memcpy(&g_emu.w->shfl_d[0][0], &v, 8);
    pthread_barrier_wait(&g_emu.w->bar);
    dp_u64 r; memcpy(&r, &g_emu.w->shfl_d[0][0], 8);
    pthread_barrier_wait(&g_emu.w->bar);
    return r;
}

DP_DEVICE int dp_lds_atomic_add(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

DP_DEVICE bool dp_any(bool p)
{
    g_emu.w->shfl[g_emu.lane][0] = p ? 1.0f : 0.0f;
    pthread_barrier_wait(&g_emu.w->bar);
    bool r = false;
    for (int i = 0; i < 64; i++) r = r || g_emu.w->shfl[i][0] != 0.0f;
    pthread_barrier_wait(&g_emu.w->bar);
    return r;
}

DP_DEVICE unsigned char* dp_lds() { return g_emu.b->lds; }
template <int BYTES>
DP_DEVICE unsigned char* dp_lds_fixed()
{
    if ((size_t)BYTES > g_emu.b->lds_bytes) __builtin_trap();
    return g_emu.b->lds;
}
/* LDS byte addresses as integers (table gathers); a read outside the workgroup's allocation returns 0 */
DP_DEVICE int dp_lds_addr(const void* p) { return (int)((const unsigned char*)p - g_emu.b->lds); }
DP_DEVICE float dp_lds_read_f32(int a)
{
    if (a < 0 || (size_t)a + 4 > g_emu.b->lds_bytes) return 0.0f;
    float v; memcpy(&v, g_emu.b->lds + a, 4); return v;
}
DP_DEVICE dpv2 dp_lds_read_v2(int a) { dpv2 v; memcpy(&v, g_emu.b->lds + a, 8); return v; }
DP_DEVICE void dp_lds_write_v2(int a, dpv2 v) { memcpy(g_emu.b->lds + a, &v, 8); }
DP_DEVICE void dp_lds_write_f32(int a, float v) { memcpy(g_emu.b->lds + a, &v, 4); }
DP_DEVICE dpf4 dp_lds_read_f4(int a) { dpf4 v; memcpy(&v, g_emu.b->lds + a, 16); return v; }
DP_DEVICE uint32_t dp_lds_read_u32(int a) { uint32_t v; memcpy(&v, g_emu.b->lds + a, 4); return v; }
DP_DEVICE uint32_t dp_lds_read_u16(int a) { uint16_t v; memcpy(&v, g_emu.b->lds + a, 2); return v; }
DP_DEVICE uint32_t dp_lds_read_u8(int a) { return g_emu.b->lds[a]; }
DP_DEVICE void dp_lds_write_u8(int a, uint32_t v) { g_emu.b->lds[a] = (unsigned char)v; }
/* (int16 half of p) * 4 + add */
DP_DEVICE int dp_mad4_i16_lo(int p, int add) { return (int)(short)(p & 0xffff) * 4 + add; }
DP_DEVICE int dp_mad4_i16_hi(int p, int add) { return (p >> 16) * 4 + add; }
DP_DEVICE int dp_pk_add_u16(int a, int b)
{
    const unsigned lo = ((unsigned)a + (unsigned)b) & 0xffffu, hi = (((unsigned)a >> 16) + ((unsigned)b >> 16)) & 0xffffu;
    return (int)(lo | (hi << 16));
}
DP_DEVICE float dp_fma(float a, float b, float c) { return fmaf(a, b, c); }
DP_DEVICE double dp_fma(double a, double b, double c) { return fma(a, b, c); }
typedef bool dp_lanes;                                 /* lane-by-lane: a lane's own membership */
#define dp_lanes_where(p) ((bool)(p))
DP_DEVICE bool dp_lanes_any(dp_lanes m) { return dp_any(m); }
DP_DEVICE void dp_sumdiff_where(dp_lanes on, dpv2& a0, dpv2& a1, dpv2& b0, dpv2& b1)
{
    if (on) {
        const dpv2 s0 = a0 + b0, s1 = a1 + b1, d0 = a0 - b0, d1 = a1 - b1;
        a0 = s0; a1 = s1; b0 = d0; b1 = d1;
    }
}
DP_DEVICE void dp_scale_where(dp_lanes on, const dpv2& l0, const dpv2& l1, float s, dpv2& r0, dpv2& r1)
{
    if (on) { r0[0] = l0[0] * s; r0[1] = l0[1] * s; r1[0] = l1[0] * s; r1[1] = l1[1] * s; }
}
DP_DEVICE dpv2 dp_fma2(dpv2 a, dpv2 b, dpv2 c) { dpv2 r; r[0] = fmaf(a[0], b[0], c[0]); r[1] = fmaf(a[1], b[1], c[1]); return r; }
DP_DEVICE void dp_store_nt(dpf4* p, dpf4 v) { *p = v; }
DP_DEVICE dpi4 dp_load_nt(const dpi4* p) { return *p; }
DP_DEVICE dpf4 dp_load_nt(const dpf4* p) { return *p; }
DP_DEVICE void dp_store2_u(float* p, float a, float b) { p[0] = a; p[1] = b; }
DP_DEVICE int dp_pcm16_pair(float a, float b)
{
    /* the device's arithmetic (devport.h), not an independent formula: clamp to [-1, 32767/32768], then ONE fused
     * multiply-add against 1.5 * 2^23, whose sum has an ulp of 1 — the rounding to nearest even is the addition's — and whose
     * low 16 mantissa bits are the integer in two's complement (tests/test_emu_kernels.py::test_int16_output checks the
     * result against rint + saturate) */
    auto one = [](float x) {
        x = fminf(fmaxf(x, -1.0f), 0.999969482421875f);
        const float f = fmaf(x, 32768.0f, 12582912.0f);
        unsigned u; memcpy(&u, &f, 4);
        return (int)(u & 0xffffu);
    };
    return one(a) | (one(b) << 16);
}
DP_DEVICE void dp_store_i2_nt(void* p, int a, int b) { memcpy(p, &a, 4); memcpy((char*)p + 4, &b, 4); }
DP_DEVICE void dp_store_i1_u(void* p, int a) { memcpy(p, &a, 4); }
DP_DEVICE unsigned long long dp_clock() { return 0; }
DP_DEVICE void dp_keep_branch() {}
DP_DEVICE void dp_sched_fence() {}

#endif
