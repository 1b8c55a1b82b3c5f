/*
 * devport.h — the handful of wavefront primitives the kernels are written against.
 *
 * Product build (hipcc, gfx950): thin wrappers over CDNA4 builtins — 64-lane wavefronts,
 * ds_bpermute cross-lane moves, wave-scope LDS fences, workgroup barrier.
 *
 * tests/emu/ provides a second implementation of exactly these names on CPU threads
 * (one thread per lane, barriers at the sync points) so the kernels' index choreography
 * can be checked in the build container, which has no GPU.  That build is test
 * infrastructure; the product library never contains it (AACG_EMU_BUILD is only defined
 * by tests/emu/Makefile).
 */
#ifndef AACG_DEVPORT_H
#define AACG_DEVPORT_H

#ifdef AACG_EMU_BUILD
#include "devport_emu.h"
#else

#include <hip/hip_runtime.h>

#define DP_DEVICE __device__ __forceinline__
#define DP_KERNEL(bounds_threads, bounds_waves) __global__ __launch_bounds__(bounds_threads, bounds_waves)

typedef float2 dpf2;
typedef float4 dpf4;
typedef double2 dpd2;
typedef int4   dpi4;
typedef uint2  dpu2;
/* (left, right) channel pair: every arithmetic op on it is one v_pk_*_f32 */
typedef float dpv2 __attribute__((ext_vector_type(2)));

DP_DEVICE int dp_tid()   { return (int)threadIdx.x; }
DP_DEVICE int dp_lane()  { return (int)(threadIdx.x & 63u); }
DP_DEVICE int dp_wave()  { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
DP_DEVICE int dp_block() { return (int)blockIdx.x; }
/* make a wave-uniform value provably uniform (lets hipcc use scalar loads behind it) */
DP_DEVICE int dp_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* LDS written by some lanes of this wave, read by others: order + complete the writes. */
DP_DEVICE void dp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DP_DEVICE void dp_block_sync() { __syncthreads(); }
/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
 * counter (vmcnt(0)), i.e. it would make every wave wait for its in-flight HBM loads; here the
 * loads keep flying across the barrier and are waited for where their data is first used. */
DP_DEVICE void dp_block_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* wave -> wave hand-off inside one workgroup through an LDS word (all waves of a workgroup are
 * resident, the producer never waits on its consumer).  Release/acquire at workgroup scope. */
DP_DEVICE void dp_flag_set(int* flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DP_DEVICE void dp_flag_wait(int* flag, int v)
{
    /* every spin is bounded (MI355X_MICROARCH.md, correctness boundaries): a producer that never arrives — which cannot
     * happen unless a wave of this workgroup died — ends the kernel with a trap after about a second instead of hanging
     * the queue */
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != v) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins == (1u << 25)) __builtin_trap();
    }
}
/* issue priority of this wave on its SIMD (0..3); s_setprio takes an immediate */
DP_DEVICE void dp_setprio(int p)
{
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

/* v[i] <- lane `src`'s v[i]  (ds_bpermute_b32; no LDS storage involved) */
template <int N>
DP_DEVICE void dp_shfl(float (&v)[N], int src)
{
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = __shfl(v[i], src, 64);
}

/* out[i] <- lane (lane ^ 7)'s in[i] without the LDS pipe: DPP row_half_mirror reverses every group of eight lanes
 * (row_mirror, lane ^ 15: every row of sixteen).  Inline assembly, eight values a block: with __builtin_amdgcn_update_dpp
 * in the run kernels hipcc 7.2's register allocator crashed under -amdgpu-sched-strategy=iterative-ilp.  Out of place (every
 * lane of the destination is written, so it needs no previous value): the in-place form of round 2 made the compiler
 * copy all the sources first where both the value and its mirror are used — 32 v_mov per channel pair.  The s_nop covers
 * the wait states a DPP read needs after a VALU write of its source (2) or — first block — of EXEC (5), which the compiler
 * cannot see from outside the block; the blocks are volatile so that they keep their order. */
#define DP_DPP_BLOCK8(NOP, CTL) \
    __asm__ volatile(NOP "\n\t" \
            "v_mov_b32_dpp %0, %8 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %9 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mov_b32_dpp %2, %10 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %11 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mov_b32_dpp %4, %12 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %13 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mov_b32_dpp %6, %14 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %15 " CTL " row_mask:0xf bank_mask:0xf" \
            : "=&v"(out[i]), "=&v"(out[i + 1]), "=&v"(out[i + 2]), "=&v"(out[i + 3]), "=&v"(out[i + 4]), "=&v"(out[i + 5]), "=&v"(out[i + 6]), "=&v"(out[i + 7]) \
            : "v"(in[i]), "v"(in[i + 1]), "v"(in[i + 2]), "v"(in[i + 3]), "v"(in[i + 4]), "v"(in[i + 5]), "v"(in[i + 6]), "v"(in[i + 7]))
template <int N>
DP_DEVICE void dp_mirror8_valu(const float (&in)[N], float (&out)[N])
{
    static_assert(N % 8 == 0, "eight values at a time");
#pragma unroll
    for (int i = 0; i < N; i += 8) { if (i == 0) DP_DPP_BLOCK8("s_nop 4", "row_half_mirror"); else DP_DPP_BLOCK8("s_nop 1", "row_half_mirror"); }
}
/* out[i] <- lane (lane ^ 15)'s in[i]: DPP row_mirror reverses every row of sixteen lanes */
template <int N>
DP_DEVICE void dp_mirror16_valu(const float (&in)[N], float (&out)[N])
{
    static_assert(N % 8 == 0, "eight values at a time");
#pragma unroll
    for (int i = 0; i < N; i += 8) { if (i == 0) DP_DPP_BLOCK8("s_nop 4", "row_mirror"); else DP_DPP_BLOCK8("s_nop 1", "row_mirror"); }
}
#undef DP_DPP_BLOCK8

/* The reorder + window step of the IMDCT for the values that come from the mirror lane (mdct.js:90-114 with
 * filter_bank.js:109-116): out[2k + c] = s_k * (mirror lane's src[2k + c]) * w[k], signs s = (-, -, -, +), as ONE
 * v_mul_f32 per value with the DPP control on its first operand — instead of a DPP move per value plus a packed
 * multiply per pair.  One rounding, like the separate multiply.  MIRROR: 16 = lane ^ 15 (row_mirror), 8 = lane ^ 7. */
template <int MIRROR>
DP_DEVICE void dp_window_mirror(const float (&src)[8], const float (&w)[4], float (&out)[8], bool first)
{
    static_assert(MIRROR == 16 || MIRROR == 8, "row_mirror or row_half_mirror");
#define DP_WM_BLOCK(NOP, CTL) \
    __asm__ volatile(NOP "\n\t" \
            "v_mul_f32_dpp %0, -%8, %16 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %1, -%9, %16 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mul_f32_dpp %2, -%10, %17 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %3, -%11, %17 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mul_f32_dpp %4, -%12, %18 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %5, -%13, %18 " CTL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_mul_f32_dpp %6, %14, %19 " CTL " row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %7, %15, %19 " CTL " row_mask:0xf bank_mask:0xf" \
            : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7]) \
            : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]), \
              "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]))
    if (MIRROR == 16) { if (first) DP_WM_BLOCK("s_nop 4", "row_mirror"); else DP_WM_BLOCK("s_nop 1", "row_mirror"); }
    else              { if (first) DP_WM_BLOCK("s_nop 4", "row_half_mirror"); else DP_WM_BLOCK("s_nop 1", "row_half_mirror"); }
#undef DP_WM_BLOCK
}

/* out[k] <- the value v of lane k of this lane's row of sixteen, k = 0..11: DPP row_newbcast on the VALU (the TNS carry's
 * state vector: twelve values held by twelve lanes of a row, wanted by every lane of the row — through LDS that was a
 * store, a wave-wide sync and three 16-byte loads on the critical path of every step) */
DP_DEVICE void dp_row_gather12(float v, float (&out)[12])
{
    __asm__ volatile("s_nop 1\n\t"
            "v_mov_b32_dpp %0, %12 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %12 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %2, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %4, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %6, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %8, %12 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %12 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %10, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %12 row_newbcast:11 row_mask:0xf bank_mask:0xf"
            : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]),
              "=&v"(out[6]), "=&v"(out[7]), "=&v"(out[8]), "=&v"(out[9]), "=&v"(out[10]), "=&v"(out[11])
            : "v"(v));
}

template <int N>
DP_DEVICE void dp_shfl(double (&v)[N], int src)
{
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = __shfl(v[i], src, 64);
}

extern __shared__ __attribute__((aligned(16))) unsigned char dp_lds_raw[];
DP_DEVICE unsigned char* dp_lds() { return dp_lds_raw; }
/* A workgroup's LDS as a static allocation of BYTES (one per kernel: a kernel must not mix this with dp_lds()): the
 * allocation then starts at a compile-time address, so table and slot offsets fold into the instructions' immediate
 * offsets instead of being added to a link-time symbol.  Such kernels are launched with no dynamic LDS. */
template <int BYTES>
DP_DEVICE unsigned char* dp_lds_fixed()
{
    /* 512-byte alignment: the run kernels XOR small offsets into absolute slot addresses and count ds_read2st64 offsets in
     * units of 512 bytes, so the block's start must not move if another LDS object ever joins it */
    __shared__ __attribute__((aligned(512))) unsigned char raw[BYTES];
    return raw;
}

/* LDS byte addresses as integers (table gathers): ds_read_b32 on a computed address.  A read outside the
 * workgroup's allocation returns 0 (the LDS has no fault path). */
DP_DEVICE int dp_lds_addr(const void* p) { return (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
DP_DEVICE float dp_lds_read_f32(int a) { return *(__attribute__((address_space(3))) const float*)(uintptr_t)(uint32_t)a; }
typedef float dp_lv4 __attribute__((ext_vector_type(4)));
DP_DEVICE dpf4 dp_lds_read_f4(int a)                   /* ds_read_b128 on a computed address */
{
    const dp_lv4 v = *(__attribute__((address_space(3))) const dp_lv4*)(uintptr_t)(uint32_t)a;
    dpf4 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; r.w = v[3]; return r;
}
/* 8-byte and 4-byte LDS accesses on computed byte addresses (the one-channel-per-wave kernels spell their addresses out as a
 * lane-dependent base, an XOR with a constant and an immediate offset) */
DP_DEVICE dpv2 dp_lds_read_v2(int a) { return *(__attribute__((address_space(3))) const dpv2*)(uintptr_t)(uint32_t)a; }
DP_DEVICE void dp_lds_write_v2(int a, dpv2 v) { *(__attribute__((address_space(3))) dpv2*)(uintptr_t)(uint32_t)a = v; }
DP_DEVICE void dp_lds_write_f32(int a, float v) { *(__attribute__((address_space(3))) float*)(uintptr_t)(uint32_t)a = v; }
DP_DEVICE uint32_t dp_lds_read_u32(int a) { return *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)(uint32_t)a; }
DP_DEVICE uint32_t dp_lds_read_u16(int a) { return *(__attribute__((address_space(3))) const uint16_t*)(uintptr_t)(uint32_t)a; }
DP_DEVICE uint32_t dp_lds_read_u8(int a) { return *(__attribute__((address_space(3))) const uint8_t*)(uintptr_t)(uint32_t)a; }
DP_DEVICE void dp_lds_write_u8(int a, uint32_t v) { *(__attribute__((address_space(3))) uint8_t*)(uintptr_t)(uint32_t)a = (uint8_t)v; }
/* (int16 half of p) * 4 + add in one VALU instruction: sign extension, scaling and the table base at once */
DP_DEVICE int dp_mad4_i16_lo(int p, int add) { int r; asm("v_mad_i32_i16 %0, %1, 4, %2" : "=v"(r) : "v"(p), "s"(add)); return r; }
DP_DEVICE int dp_mad4_i16_hi(int p, int add) { int r; asm("v_mad_i32_i16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(p), "s"(add)); return r; }
typedef unsigned short dp_u16x2 __attribute__((ext_vector_type(2)));
DP_DEVICE int dp_pk_add_u16(int a, int b)
{
    return __builtin_bit_cast(int, (dp_u16x2)(__builtin_bit_cast(dp_u16x2, a) + __builtin_bit_cast(dp_u16x2, b)));
}
DP_DEVICE float dp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DP_DEVICE double dp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
DP_DEVICE dpv2 dp_fma2(dpv2 a, dpv2 b, dpv2 c) { return __builtin_elementwise_fma(a, b, c); }
/* ---- one complex value per register pair: (re, im) as a dpv2, every operation one v_pk_*_f32 ---- */
DP_DEVICE dpv2 dp_cswap(dpv2 a) { return __builtin_shufflevector(a, a, 1, 0); }
/* a + i b and a - i b: the swap is the instruction's op_sel, the signs a constant pair — one v_pk_fma_f32, exact in the
 * product (x 1 / x -1), rounded once like the addition it stands for */
DP_DEVICE dpv2 dp_cadd_i(dpv2 a, dpv2 b) { const dpv2 k = {-1.0f, 1.0f}; return __builtin_elementwise_fma(dp_cswap(b), k, a); }
DP_DEVICE dpv2 dp_csub_i(dpv2 a, dpv2 b) { const dpv2 k = {1.0f, -1.0f}; return __builtin_elementwise_fma(dp_cswap(b), k, a); }
/* a * w (complex): (a.re w.re, a.im w.re), then (-a.im w.im, a.re w.im) on top — two instructions; left to the compiler the
 * (-w.im, w.im) pair was built with a v_xor + v_mov per multiply */
DP_DEVICE dpv2 dp_cmul(dpv2 a, dpv2 w)
{
    dpv2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}

/* wave -> wave hand-off by phase: wait until the flag has reached at least v (flags only ever grow within a launch) */
DP_DEVICE void dp_flag_wait_ge(int* flag, int v)
{
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins == (1u << 25)) __builtin_trap();
    }
}

/* ---- workgroup -> workgroup through global memory (the rendezvous cells of the _rv run kernels) ---------------------
 * MI355X_MICROARCH.md, inter-workgroup visibility: payload as agent-scope (sc1, write-through) 8-byte stores, drained with
 * s_waitcnt vmcnt(0) BEFORE the state word changes; the reader takes the state word with an agent-scope load and the
 * payload with agent-scope (sc1) loads, which bypass its CU's L1.  No fences (an agent-scope release is a write-back of the
 * whole L2), no assumption on dispatch order or on which XCD either side runs. */
typedef unsigned long long dp_u64;
DP_DEVICE dp_u64 dp_g_load_u64(const dp_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DP_DEVICE bool dp_g_cas_u64(dp_u64* p, dp_u64 expected, dp_u64 desired)
{
    return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DP_DEVICE void dp_g_store_f2(float* p, float a, float b)
{
    dpf2 v; v.x = a; v.y = b;
    __hip_atomic_store((dp_u64*)p, __builtin_bit_cast(dp_u64, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DP_DEVICE dpf2 dp_g_load_f2(const float* p)
{
    return __builtin_bit_cast(dpf2, __hip_atomic_load((const dp_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
DP_DEVICE float dp_g_load_f1(const float* p)
{
    return __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
DP_DEVICE void dp_g_store_u64(dp_u64* p, dp_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DP_DEVICE unsigned dp_g_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DP_DEVICE void dp_g_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DP_DEVICE void dp_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
/* lane 0's value in every lane (a 64-bit scalar) */
DP_DEVICE dp_u64 dp_first_u64(dp_u64 v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((dp_u64)hi << 32) | lo;
}

/* which CU of the chip this wave runs on (profiling): XCC_ID[3:0] above HW_ID's SE_ID / SH_ID / CU_ID fields (bits 14:8) */
DP_DEVICE unsigned dp_cu_id()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 15u) << 7) | ((hw >> 8) & 0x7fu);
}
/* Scalar loads issued where the SOURCE says, and waited for where the source says.  hipcc turns uniform reads of read-only
 * memory into s_load by itself, but it decides the order: a prologue's loads come out as a chain — load, wait, use, next load —
 * even where they are independent (every result's first use gets its own s_waitcnt lgkmcnt(0), and loads are not hoisted above
 * it).  A run kernel's start is nothing BUT dependent round trips (aacg_run, aacg_device.h), so there the batch is spelled
 * out: dp_sload*() issue, dp_swait() is the one wait, and its in/out operands keep every use of the results behind it.
 * No memory clobber anywhere: that would turn the compiler's own later uniform reads into vector loads. */
typedef unsigned dp_su2 __attribute__((ext_vector_type(2)));
typedef unsigned dp_su4 __attribute__((ext_vector_type(4)));
typedef unsigned dp_su8 __attribute__((ext_vector_type(8)));
/* (not `volatile`, which would count as a clobber of memory — see above; their outputs feed dp_swait, which keeps them in place) */
DP_DEVICE unsigned dp_sload1(const void* p, int byte_off) { unsigned r; asm("s_load_dword %0, %1, %2" : "=s"(r) : "s"(p), "s"(byte_off)); return r; }
DP_DEVICE dp_su4 dp_sload4(const void* p) { dp_su4 r; asm("s_load_dwordx4 %0, %1, 0x0" : "=s"(r) : "s"(p)); return r; }
DP_DEVICE dp_su8 dp_sload8(const void* p) { dp_su8 r; asm("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(p)); return r; }
DP_DEVICE void dp_swait(dp_su8& a, dp_su4& b, unsigned& c, unsigned& d, unsigned& e)
{
    asm("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e));
}
/* `also`: a pointer the batch does not read but that must have ARRIVED by now — the compiler fetches kernel arguments lazily,
 * one first used behind a batch would be fetched behind it, a round trip later */
DP_DEVICE dp_su8 dp_sload8(const void* p, const void* also) { dp_su8 r; asm("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(p), "s"(also)); return r; }

/* true in every lane if the predicate holds in any lane of the wave */
/* LDS bump allocation: returns the old value */
DP_DEVICE int dp_lds_atomic_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

DP_DEVICE bool dp_any(bool p) { return __any(p) != 0; }

/* A set of lanes as a value: the wave's lane mask in a scalar register pair (what a compare produces anyway). */
typedef unsigned long long dp_lanes;
#define dp_lanes_where(p) __builtin_amdgcn_ballot_w64(p)
DP_DEVICE bool dp_lanes_any(dp_lanes m) { return m != 0; }

/* Lanes of m: (a, b) <- (a + b, a - b) on two register pairs, the other lanes untouched — the additions run under
 * an EXEC mask instead of being computed everywhere and selected per element (2 v_pk_add + 4 v_cndmask per pair become
 * 2 v_pk_add + 1 v_pk_mov). */
DP_DEVICE void dp_sumdiff_where(dp_lanes m, dpv2& a0, dpv2& a1, dpv2& b0, dpv2& b1)
{
    unsigned long long sv;
    dpv2 t0, t1;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "v_pk_add_f32 %[t0], %[a0], %[b0]\n\t"
                 "v_pk_add_f32 %[t1], %[a1], %[b1]\n\t"
                 "v_pk_add_f32 %[b0], %[a0], %[b0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %[b1], %[a1], %[b1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mov_b32 %[a0], %[t0], %[t0] op_sel:[0,1]\n\t"
                 "v_pk_mov_b32 %[a1], %[t1], %[t1] op_sel:[0,1]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [t0] "=&v"(t0), [t1] "=&v"(t1), [a0] "+v"(a0), [a1] "+v"(a1), [b0] "+v"(b0), [b1] "+v"(b1)
                 : [m] "s"(m)
                 : "scc");
}
/* Lanes of m: r <- l * s on two register pairs (v_pk_mul_f32 under an EXEC mask), the other lanes untouched. */
DP_DEVICE void dp_scale_where(dp_lanes m, const dpv2& l0, const dpv2& l1, float s, dpv2& r0, dpv2& r1)
{
    unsigned long long sv;
    dpv2 sc; sc[0] = sc[1] = s;
    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                 "v_pk_mul_f32 %[r0], %[l0], %[sc]\n\t"
                 "v_pk_mul_f32 %[r1], %[l1], %[sc]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [sv] "=&s"(sv), [r0] "+v"(r0), [r1] "+v"(r1)
                 : [m] "s"(m), [l0] "v"(l0), [l1] "v"(l1), [sc] "v"(sc)
                 : "scc");
}
/* streaming (non-temporal) 16-byte accesses: PCM is written once and never re-read by the kernel,
 * spectra are read once */
typedef float dp_nv4 __attribute__((ext_vector_type(4)));
/* streaming store of 16 bytes (global_store_dwordx4 ... nt).  Measured against the alternatives on the PCM stores:
 * plain 23.0 us, nt 19.2 us; sc0 sc1 / sc1 (write-through scopes) 18.8-18.9 us against nt 17.0 at that time;
 * sc0 sc1 nt and sc1 nt no better than nt alone. */
DP_DEVICE void dp_store_nt(dpf4* p, dpf4 v)
{
    dp_nv4 t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    __builtin_nontemporal_store(t, (dp_nv4*)p);
}
/* streaming loads of 16 bytes (global_load_dwordx4 ... nt): the spectra of a multichannel batch.  There the PCM goes out as the
 * elements' partial lines, which the L2 merges — its write requests bound those batches — and read-once spectra that do not
 * allocate in it leave it more room for that (config 5: 64.5 -> 62.0 us with the hint, f32 seam 69.3 -> 65.6; HBM traffic
 * unchanged).  Stereo batches store whole lines and are 4 % slower with it (12.1 -> 12.6 us), so the hint is a kernel
 * variant, not the default. */
typedef int dp_nv4i __attribute__((ext_vector_type(4)));
DP_DEVICE dpi4 dp_load_nt(const dpi4* p)
{
    const dp_nv4i t = __builtin_nontemporal_load((const dp_nv4i*)p);
    dpi4 r; r.x = t[0]; r.y = t[1]; r.z = t[2]; r.w = t[3]; return r;
}
DP_DEVICE dpf4 dp_load_nt(const dpf4* p)
{
    const dp_nv4 t = __builtin_nontemporal_load((const dp_nv4*)p);
    dpf4 r; r.x = t[0]; r.y = t[1]; r.z = t[2]; r.w = t[3]; return r;
}
/* two adjacent floats at an address that is only 4-byte aligned (odd channel counts): one
 * global_store_dwordx2 — gfx950 runs in unaligned-access mode, and the type says align 4 */
typedef float dp_f2u __attribute__((ext_vector_type(2), aligned(4)));
DP_DEVICE void dp_store2_u(float* p, float a, float b) { dp_f2u v; v[0] = a; v[1] = b; *(dp_f2u*)p = v; }
/* two PCM-scale samples (|x| < 1 nominally) as int16, round to nearest even, saturating: low half = a */
typedef short dp_i16x2 __attribute__((ext_vector_type(2)));
DP_DEVICE int dp_pcm16_pair(float a, float b)
{
    /* clamp to the int16 range, then x * 32768 + 1.5 * 2^23 in ONE fused multiply-add: the sum has an ulp of 1, so the
     * rounding to nearest even is the addition's, and the low 16 mantissa bits are the integer in two's complement;
     * v_perm_b32 packs the two low halves.  Five instructions per pair (multiply, round, convert twice, pack: seven). */
    a = __builtin_amdgcn_fmed3f(a, -1.0f, 0.999969482421875f);
    b = __builtin_amdgcn_fmed3f(b, -1.0f, 0.999969482421875f);
    const unsigned ua = __builtin_bit_cast(unsigned, __builtin_fmaf(a, 32768.0f, 12582912.0f));
    const unsigned ub = __builtin_bit_cast(unsigned, __builtin_fmaf(b, 32768.0f, 12582912.0f));
    return (int)__builtin_amdgcn_perm(ub, ua, 0x05040100u);
}
typedef int dp_i2u __attribute__((ext_vector_type(2), aligned(4)));
DP_DEVICE void dp_store_i2_nt(void* p, int a, int b) { dp_i2u v; v[0] = a; v[1] = b; __builtin_nontemporal_store(v, (dp_i2u*)p); }
typedef int dp_i1u __attribute__((aligned(2)));
DP_DEVICE void dp_store_i1_u(void* p, int a) { *(dp_i1u*)p = a; }
/* constant-rate (100 MHz) wall clock, same time base on every CU: phase timelines for profiling */
DP_DEVICE unsigned long long dp_clock() { return wall_clock64(); }
/* keep the instruction scheduler from hoisting the next block's loads above this point
 * (bounds the number of gathers in flight, i.e. VGPR pressure) */
/* inside a block guarded by a wave-uniform condition: keeps it a real (scalar) branch */
DP_DEVICE void dp_keep_branch() { asm volatile(""); }
DP_DEVICE void dp_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

#endif /* AACG_EMU_BUILD */
#endif
