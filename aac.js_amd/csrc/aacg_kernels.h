/*
 * aacg_kernels.h — the hot path as wavefront code for CDNA4 (gfx950).
 *
 * Work decomposition:
 *   wave       = one unit (SCE/LFE/CPE) of one frame: both channels of a CPE live in one
 *                wave, so MS/IS are register-local and stereo PCM leaves as 16-byte
 *                (L,R,L,R) stores, 1 KiB contiguous per wave instruction.
 *   workgroup  = 16 waves = one run: consecutive frames of one element of one stream, one
 *                workgroup per CU (128 VGPRs per lane, ~158 KiB of LDS).  Tails travel
 *                wave -> wave through LDS, released by one LDS flag per wave (no workgroup
 *                barrier on the hand-off); the first frame of a chain starts from the overlap
 *                state in HBM, a later run recomputes the tail of the frame before it (a wave
 *                of its own, or double duty of its first wave; no inter-workgroup communication).
 *   optional   = AACG_TNS_SPEC filters and AACG_PNS_SPEC noise bands run in a kernel of their own
 *                (spectral_ex_body) that hands f32 spectra to the f32 run kernel: the run kernels
 *                never carry them.
 *   tables     = rotation/twiddle/window (and dequant) tables are copied into LDS once per
 *                workgroup; per-wave table reads never use the vector-memory pipeline.
 *   HBM        = spectra in with 16-byte loads in natural order, redistributed to the FFT
 *                lane map through LDS; PCM out with 16-byte stores.
 *   IMDCT      = N/4-point complex inverse FFT between two rotations (the algorithm class of
 *                mdct.js:62-115) as radix-8 register butterflies with two LDS transposes
 *                (512 = 8x8x8) or one (64 = 8x8 per short window); natural order in and out,
 *                so no bit-reversal pass (fft.js:113-137) exists.
 *
 * Lane maps.  Long: lane l, element j holds index l + 64 j.  Short: lane (w = l>>3, g = l&7),
 * element j holds index g + 8 j of window w.
 */
#ifndef AACG_KERNELS_H
#define AACG_KERNELS_H

#include "devport.h"
#include "aacg_device.h"

/* Work-skipping / tracing switches for tools/ (timeline, ablations).  They exist only in a build made with
 * -DAACG_PROFILE (`make profile` -> variants/profile.so): the library that ships has no such branches and ignores
 * AACG_ABLATE.  Bits: 1 skip IMDCT (f32 seam), 2 skip PCM stores, 8 skip dequant arithmetic, 16 per-wave phase
 * timestamps, 32 / 64 priority schemes, 128 no load stagger. */
#ifdef AACG_PROFILE
#define AACG_ABL(P, bits) ((P).ablate & (bits))
#else
#define AACG_ABL(P, bits) 0
#endif

struct cpx { float re, im; };

DP_DEVICE cpx c_add(cpx a, cpx b) { cpx r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
DP_DEVICE cpx c_sub(cpx a, cpx b) { cpx r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
DP_DEVICE cpx c_muli(cpx a)       { cpx r; r.re = -a.im; r.im = a.re; return r; }            /* i * a */
DP_DEVICE cpx c_mul(cpx a, cpx w)
{
#pragma clang fp contract(off)
    cpx r;
    r.re = dp_fma(a.re, w.re, -(a.im * w.im));
    r.im = dp_fma(a.re, w.im, a.im * w.re);
    return r;
}

/* 8-point inverse DFT  y[q] = sum_j x[j] e^{+2 pi i j q / 8}, in place. */
DP_DEVICE void radix8_inv(cpx (&x)[8])
{
    /* the single-channel (planar) transform is instantiated in several kernels of different translation units (plain, coupling,
     * optional stages): with multiply-adds formed wherever the compiler finds a product next to a sum, those kernels disagreed
     * in the last bit on the GPU (the fused and the staged coupling routes on single-channel elements).  Every multiply-add of
     * this path is spelled out (dp_fma); nothing else is fused. */
#pragma clang fp contract(off)
    const float h = 0.70710678118654752440f;
    cpx a0 = c_add(x[0], x[4]), a1 = c_sub(x[0], x[4]);
    cpx a2 = c_add(x[2], x[6]), a3 = c_sub(x[2], x[6]);
    cpx a4 = c_add(x[1], x[5]), a5 = c_sub(x[1], x[5]);
    cpx a6 = c_add(x[3], x[7]), a7 = c_sub(x[3], x[7]);
    cpx e0 = c_add(a0, a2), e2 = c_sub(a0, a2);
    cpx ia3 = c_muli(a3), ia7 = c_muli(a7);
    cpx e1 = c_add(a1, ia3), e3 = c_sub(a1, ia3);
    cpx o0 = c_add(a4, a6), o2 = c_sub(a4, a6);
    cpx o1 = c_add(a5, ia7), o3 = c_sub(a5, ia7);
    cpx t1, t2, t3;
    t1.re = (o1.re - o1.im) * h;  t1.im = (o1.re + o1.im) * h;     /* o1 * (1+i)/sqrt2  */
    t2 = c_muli(o2);                                               /* o2 * i            */
    t3.re = -(o3.re + o3.im) * h; t3.im = (o3.re - o3.im) * h;     /* o3 * (-1+i)/sqrt2 */
    x[0] = c_add(e0, o0); x[4] = c_sub(e0, o0);
    x[1] = c_add(e1, t1); x[5] = c_sub(e1, t1);
    x[2] = c_add(e2, t2); x[6] = c_sub(e2, t2);
    x[3] = c_add(e3, t3); x[7] = c_sub(e3, t3);
}

/* ---- one complex value per register pair, (re, im) packed: the single-channel transforms below — a butterfly is v_pk_add /
 * v_pk_fma with op_sel swaps, a rotation two packed instructions ---- */
DP_DEVICE dpv2 k8_ld2(const float* p) { const dpf2 t = *(const dpf2*)p; dpv2 r; r[0] = t.x; r[1] = t.y; return r; }
DP_DEVICE void k8_st2(float* p, dpv2 v) { dpf2 t; t.x = v[0]; t.y = v[1]; *(dpf2*)p = t; }
DP_DEVICE dpv2 k8_v2(float a, float b) { dpv2 r; r[0] = a; r[1] = b; return r; }

/* 8-point inverse DFT on (re, im) pairs: 26 packed operations */
DP_DEVICE void k8_radix8(dpv2 (&x)[8])
{
    const dpv2 h = {0.70710678118654752440f, 0.70710678118654752440f};
    const dpv2 mh = {-0.70710678118654752440f, -0.70710678118654752440f};
    const dpv2 a0 = x[0] + x[4], a1 = x[0] - x[4], a2 = x[2] + x[6], a3 = x[2] - x[6];
    const dpv2 a4 = x[1] + x[5], a5 = x[1] - x[5], a6 = x[3] + x[7], a7 = x[3] - x[7];
    const dpv2 e0 = a0 + a2, e2 = a0 - a2, e1 = dp_cadd_i(a1, a3), e3 = dp_csub_i(a1, a3);
    const dpv2 o0 = a4 + a6, o2 = a4 - a6, o1 = dp_cadd_i(a5, a7), o3 = dp_csub_i(a5, a7);
    const dpv2 s1 = dp_cadd_i(o1, o1);                 /* o1 (1 + i)  */
    const dpv2 s3 = dp_cadd_i(-o3, o3);                /* o3 (-1 + i) */
    x[0] = e0 + o0; x[4] = e0 - o0;
    x[1] = dp_fma2(s1, h, e1); x[5] = dp_fma2(s1, mh, e1);
    x[2] = dp_cadd_i(e2, o2); x[6] = dp_csub_i(e2, o2);
    x[3] = dp_fma2(s3, h, e3); x[7] = dp_fma2(s3, mh, e3);
}

DP_DEVICE void lds_put(float* base, int idx, cpx v)
{
    dpf2 t; t.x = v.re; t.y = v.im;
    *(dpf2*)(base + 2 * idx) = t;
}
DP_DEVICE cpx lds_get(const float* base, int idx)
{
    dpf2 t = *(const dpf2*)(base + 2 * idx);
    cpx v; v.re = t.x; v.im = t.y; return v;
}

/* ------------------------------------------------------------------------------------ */
/* unit descriptor: fetched as 16 scalar dwords (the address is wave-uniform), unpacked with    */
/* SALU bit operations — byte-wide field reads would each be a vector-memory round trip          */
/* ------------------------------------------------------------------------------------ */
struct unit_view {
    uint32_t pcm_offset, coef_offset, meta_offset, tns_offset;
    int channel, n_out_ch, n_ch, flags;
    int seq[2], shape[2], shape_prev[2], max_sfb[2], tns[2];
    uint32_t gmap[2];           /* 4 bits per window: its group (planner-filled) */
    uint32_t cpl_first, cpl_n;  /* AACG_CCE_SPEC: this unit's independent-coupling jobs (aacg_dev_unit) */
};
DP_DEVICE unit_view load_unit(const aacg_dev_unit* u)
{
    const uint32_t* w = (const uint32_t*)u;
    unit_view v;
    const uint32_t w2 = w[2], w3 = w[3];
    v.pcm_offset = w[1]; v.coef_offset = w[4]; v.meta_offset = w[5];
    v.channel = (int)(w2 & 0xffffu); v.n_out_ch = (int)(w2 >> 16);
    v.n_ch = (int)(w3 & 0xffu); v.flags = (int)((w3 >> 8) & 0xffu);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t ci = w[6 + 4 * c], cj = w[7 + 4 * c];
        v.seq[c] = (int)(ci & 0xffu); v.shape[c] = (int)((ci >> 8) & 0xffu);
        v.shape_prev[c] = (int)((ci >> 16) & 0xffu); v.max_sfb[c] = (int)(ci >> 24);
        v.tns[c] = (int)((cj >> 8) & AACG_CHAN_TNS_PRESENT);         /* aacg_chan_info.flags */
        v.gmap[c] = w[16 + c];
    }
    v.tns_offset = w[14];
    v.cpl_first = w[18]; v.cpl_n = w[19];
    return v;
}

/* ------------------------------------------------------------------------------------ */
/* table staging: global (L2) -> LDS, once per workgroup                                   */
/* ------------------------------------------------------------------------------------ */
/* Split in two so that the table loads are issued BEFORE the wave's own spectrum loads: vector
 * loads return in order, so the LDS copy (which waits for the table data only) does not wait for
 * the spectrum, and the spectrum keeps flying across the workgroup barrier. */
DP_DEVICE void stage_tables_load(const aacg_tables* T, int n_floats, dpf4& t0, dpf4& t1)
{
    const dpf4* src = (const dpf4*)T;
    const int n4 = n_floats >> 2, tid = dp_tid();
    const int i1 = tid + AACG_WG_THREADS;
    t0 = src[tid < n4 ? tid : n4 - 1];                 /* clamped: unconditional loads, conditional stores */
    t1 = src[i1 < n4 ? i1 : n4 - 1];
}
DP_DEVICE void stage_tables_store(float* lds, int n_floats, const dpf4& t0, const dpf4& t1)
{
    dpf4* dst = (dpf4*)lds;
    const int n4 = n_floats >> 2, tid = dp_tid();
    if (tid < n4) dst[tid] = t0;
    if (tid + AACG_WG_THREADS < n4) dst[tid + AACG_WG_THREADS] = t1;
}
DP_DEVICE void stage_tables(const aacg_tables* T, float* lds, int n_floats)
{
    dpf4 t0, t1;
    stage_tables_load(T, n_floats, t0, t1);
    stage_tables_store(lds, n_floats, t0, t1);
}

/* ------------------------------------------------------------------------------------ */
/* windows of the long sequences, filter_bank.js:105-141,180-202                           */
/* ------------------------------------------------------------------------------------ */
/* (w[n], w[n+1]) multiplying IMDCT output n, n+1 of the first half (n even). */
DP_DEVICE dpf2 head_window(const float* tab, int seq, int shape_prev, int n)
{
    const float* wl = tab + AACG_TAB_OFF_WIN_LONG + 1024 * shape_prev;
    if (seq != AACG_LONG_STOP_SEQUENCE) return *(const dpf2*)(wl + n);      /* filter_bank.js:109-111,124-126 */
    /* LONG_STOP: 0 | previous-shape short window | 1   (filter_bank.js:185-195) */
    const float* ws = tab + AACG_TAB_OFF_WIN_SHORT + 128 * shape_prev;
    int i = n - 448; i = i < 0 ? 0 : (i > 126 ? 126 : i);
    dpf2 v = *(const dpf2*)(ws + i);
    if (n < 448) { v.x = 0.0f; v.y = 0.0f; }
    if (n >= 576) { v.x = AACG_PCM_SCALE; v.y = AACG_PCM_SCALE; }     /* "1": the tables carry the PCM scale */
    return v;
}
/* (w[n], w[n+1]) multiplying IMDCT output 1024 + n, 1024 + n + 1 (n even). */
DP_DEVICE dpf2 tail_window(const float* tab, int seq, int shape, int n)
{
    dpf2 v, r;
    if (seq != AACG_LONG_START_SEQUENCE) {                                   /* reversed long window, filter_bank.js:114-116 */
        v = *(const dpf2*)(tab + AACG_TAB_OFF_WIN_LONG + 1024 * shape + 1022 - n);
        r.x = v.y; r.y = v.x;
        return r;
    }
    /* LONG_START: 1 | reversed short window | 0   (filter_bank.js:129-139) */
    int i = 574 - n; i = i < 0 ? 0 : (i > 126 ? 126 : i);
    v = *(const dpf2*)(tab + AACG_TAB_OFF_WIN_SHORT + 128 * shape + i);
    r.x = v.y; r.y = v.x;
    if (n < 448) { r.x = AACG_PCM_SCALE; r.y = AACG_PCM_SCALE; }
    if (n >= 576) { r.x = 0.0f; r.y = 0.0f; }
    return r;
}

/* x * w where a window value of exactly 0 means "this region takes nothing from the IMDCT output"
 * (LONG_STOP first 448 samples, LONG_START last 448): the reference copies / zeroes there instead of
 * multiplying (filter_bank.js:129-139,185-188), so a NaN or Inf in the spectrum must not leak through.
 * Real window samples are never 0, so the test is exact; `guard` is wave-uniform (only START / STOP). */
/* The products are formed unconditionally; the fix-up sits behind a wave-uniform branch that only START / STOP
 * frames take (dp_keep_branch() stops the compiler from turning it back into per-element selects). */
DP_DEVICE void wfix(float& r, float w) { r = (w == 0.0f) ? 0.0f : r; }

/* ------------------------------------------------------------------------------------ */
/* LDS transposes between radix-8 stages: 512 complex per channel, XOR-swizzled so that the   */
/* 8-byte writes (16-lane groups, 16 slots) and reads (32-lane groups, 32 slots) are both    */
/* bank-conflict-free without padding                                                     */
/* ------------------------------------------------------------------------------------ */
DP_DEVICE int xch1(int q, int l)     { return 64 * q + (l ^ (q << 3)); }                 /* row q (0..7) of 64 */
DP_DEVICE int xch2(int row, int col) { return 8 * row + (col ^ ((row >> 2) & 7)); }      /* row 0..63 of 8     */

struct chan_par { int seq, shape, shape_prev; };

/* ------------------------------------------------------------------------------------ */
/* long windows: IMDCT-2048 + window, mdct.js:62-115 + filter_bank.js                      */
/* ------------------------------------------------------------------------------------ */
DP_DEVICE int long_col(int l);

/* NC channels advance together as independent instruction streams (ILP across the LDS and
 * FMA latencies); they share the rotation and twiddle reads.  area[c][0..1023] holds channel
 * c's spectrum in natural order on entry, is reused for its FFT transposes and receives its
 * windowed second half (natural order).  Returns the windowed first half at n = 2 l + 128 m
 * (hx[c][m]) and n + 1 (hy[c][m]). */
/* window + reorder of one planar long channel for one window sequence; m[r] = re, m[8+r] = im of lane 63 - l */
template <int SEQ>
DP_DEVICE void long_planar_window(const float* tab, const chan_par& cp, bool want_head, float* area, int l /* column */,
                                  const float (&R)[8], const float (&I)[8], const float (&m)[16],
                                  float (&hx)[8], float (&hy)[8])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = 2 * l + 128 * j;
        const dpf2 w0 = head_window(tab, SEQ, cp.shape_prev, n), w1 = head_window(tab, SEQ, cp.shape_prev, n + 512);
        const dpf2 v0 = tail_window(tab, SEQ, cp.shape, n), v1 = tail_window(tab, SEQ, cp.shape, n + 512);
        if (want_head) {
            hx[j]     = I[j + 4] * w0.x;              /* y[2k]        =  im[N/8 + k]     */
            hy[j]     = -m[3 - j] * w0.y;             /* y[2k+1]      = -re[N/8 - 1 - k] */
            hx[j + 4] = R[j] * w1.x;                  /* y[N/4+2k]    =  re[k]           */
            hy[j + 4] = -m[8 + 7 - j] * w1.y;         /* y[N/4+2k+1]  = -im[N/4 - 1 - k] */
            if (SEQ == AACG_LONG_STOP_SEQUENCE) { wfix(hx[j], w0.x); wfix(hy[j], w0.y); wfix(hx[j + 4], w1.x); wfix(hy[j + 4], w1.y); }
        }
        dpf2 t, t2;
        t.x = R[j + 4] * v0.x;                        /* y[N/2+2k]    =  re[N/8 + k]     */
        t.y = -m[8 + 3 - j] * v0.y;                   /* y[N/2+2k+1]  = -im[N/8 - 1 - k] */
        t2.x = -I[j] * v1.x;                          /* y[3N/4+2k]   = -im[k]           */
        t2.y = m[7 - j] * v1.y;                       /* y[3N/4+2k+1] =  re[N/4 - 1 - k] */
        if (SEQ == AACG_LONG_START_SEQUENCE) { wfix(t.x, v0.x); wfix(t.y, v0.y); wfix(t2.x, v1.x); wfix(t2.y, v1.y); }
        *(dpf2*)(area + n) = t;
        *(dpf2*)(area + n + 512) = t2;
    }
}

template <int NC, bool VM = false>                     /* VM: columns dealt out by long_col (below), mirror exchange as DPP row_mirror */
DP_DEVICE void long_channels(const float* tab, const chan_par (&cp)[NC], bool want_head,
                             float* const (&area)[NC], float (&hx)[NC][8], float (&hy)[NC][8])
{
#pragma clang fp contract(off)
    const int l = dp_lane();
    const float* sincos = tab + AACG_TAB_OFF_SINCOS_LONG;

    /* k = l + 64 j:  X[2k] and X[N/2-1-2k] (mdct.js:74-75).  One complex value per register pair (round 4): a single channel's
     * transform in half the vector instructions of the scalar form it replaces */
    dpv2 z[NC][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const dpv2 sc = k8_ld2(sincos + 2 * (64 * j + l));
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const dpf2 a = *(const dpf2*)(area[c] + 2 * l + 128 * j);
            const dpf2 b = *(const dpf2*)(area[c] + 1022 - 2 * l - 128 * j);
            z[c][j] = dp_cmul(k8_v2(b.y, a.x), sc);                  /* (re, im) = (b c - a s, a c + b s): mdct.js:74-75 */
        }
    }
    dp_wave_sync();

    /* 512-point inverse FFT, unscaled (fft.js with forward = false) */
#pragma unroll
    for (int c = 0; c < NC; c++) k8_radix8(z[c]);      /* over j (stride 64)       */
#pragma unroll
    for (int q = 1; q < 8; q++) {
        const dpv2 tw = k8_ld2(tab + AACG_TAB_OFF_TW512 + 2 * (64 * (q - 1) + l));
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][q] = dp_cmul(z[c][q], tw);
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int c = 0; c < NC; c++) k8_st2(area[c] + 2 * xch1(q, l), z[c][q]);
    dp_wave_sync();
    const int l0 = l & 7, qq = l >> 3;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][j] = k8_ld2(area[c] + 2 * xch1(qq, l0 + 8 * j));
    dp_wave_sync();
#pragma unroll
    for (int c = 0; c < NC; c++) k8_radix8(z[c]);      /* over l1 (stride 8)       */
#pragma unroll
    for (int r = 1; r < 8; r++) {
        const dpv2 tw = k8_ld2(tab + AACG_TAB_OFF_TW64 + 2 * (8 * (r - 1) + l0));
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][r] = dp_cmul(z[c][r], tw);
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < NC; c++) k8_st2(area[c] + 2 * xch2(qq + 8 * r, l0), z[c][r]);
    dp_wave_sync();
    const int col = VM ? long_col(l) : l;              /* this lane's column from here on */
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][i] = k8_ld2(area[c] + 2 * xch2(col, i));
    dp_wave_sync();
#pragma unroll
    for (int c = 0; c < NC; c++) k8_radix8(z[c]);      /* over l0; the lane now holds Z[col + 64 r] */

    /* post-IFFT rotation (mdct.js:82-87) */
    float R[NC][8], I[NC][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const dpv2 sc = k8_ld2(sincos + 2 * (64 * r + col));
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const dpv2 ri = dp_cmul(z[c][r], sc);
            R[c][r] = ri[0]; I[c][r] = ri[1];
        }
    }

#pragma unroll
    for (int c = 0; c < NC; c++) {
        /* the mirror lane's values for the reorder: m[r] = re, m[8+r] = im of lane 63 - l */
        float m[16], own[16];
#pragma unroll
        for (int r = 0; r < 8; r++) { own[r] = m[r] = R[c][r]; own[8 + r] = m[8 + r] = I[c][r]; }
        if (VM) dp_mirror16_valu(own, m); else dp_shfl(m, 63 - l);
        /* reorder (mdct.js:90-114) fused with the window (filter_bank.js:109-116 etc.), with the sequence as a
         * compile-time constant: the branches inside head_window / tail_window fold, the reads go out together */
        if (cp[c].seq == AACG_ONLY_LONG_SEQUENCE)       long_planar_window<AACG_ONLY_LONG_SEQUENCE>(tab, cp[c], want_head, area[c], col, R[c], I[c], m, hx[c], hy[c]);
        else if (cp[c].seq == AACG_LONG_START_SEQUENCE) long_planar_window<AACG_LONG_START_SEQUENCE>(tab, cp[c], want_head, area[c], col, R[c], I[c], m, hx[c], hy[c]);
        else                                            long_planar_window<AACG_LONG_STOP_SEQUENCE>(tab, cp[c], want_head, area[c], col, R[c], I[c], m, hx[c], hy[c]);
    }
}
template <int NC, bool VM = false>                     /* VM: the l ^ 7 exchange on the VALU (dp_mirror8_valu) */
DP_DEVICE void short_channels(const float* tab, const chan_par (&cp)[NC],
                              float* const (&area)[NC], float (&hx)[NC][8], float (&hy)[NC][8])
{
#pragma clang fp contract(off)
    const int l = dp_lane(), w = l >> 3, g = l & 7;
    const float* sincos = tab + AACG_TAB_OFF_SINCOS_SHORT;

    dpv2 z[NC][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const dpv2 sc = k8_ld2(sincos + 2 * (8 * j + g));
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const dpf2 a = *(const dpf2*)(area[c] + 128 * w + 2 * g + 16 * j);          /* X_w[2k], k = g + 8 j */
            const dpf2 b = *(const dpf2*)(area[c] + 128 * w + 126 - 2 * g - 16 * j);    /* .y = X_w[127 - 2k]   */
            z[c][j] = dp_cmul(k8_v2(b.y, a.x), sc);
        }
    }
    dp_wave_sync();

    /* 64-point inverse FFT per window: 8 lanes x 8 points */
#pragma unroll
    for (int c = 0; c < NC; c++) k8_radix8(z[c]);
#pragma unroll
    for (int q = 1; q < 8; q++) {
        const dpv2 tw = k8_ld2(tab + AACG_TAB_OFF_TW64 + 2 * (8 * (q - 1) + g));
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][q] = dp_cmul(z[c][q], tw);
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
        for (int c = 0; c < NC; c++) k8_st2(area[c] + 2 * xch2(8 * w + q, g), z[c][q]);
    dp_wave_sync();
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < NC; c++) z[c][i] = k8_ld2(area[c] + 2 * xch2(l, i));
    dp_wave_sync();
#pragma unroll
    for (int c = 0; c < NC; c++) k8_radix8(z[c]);      /* lane (w, q) holds Z_w[q + 8 r] */

    float R[NC][8], I[NC][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const dpv2 sc = k8_ld2(sincos + 2 * (8 * r + g));
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const dpv2 ri = dp_cmul(z[c][r], sc);
            R[c][r] = ri[0]; I[c][r] = ri[1];
        }
    }

#pragma unroll
    for (int c = 0; c < NC; c++) {
        float m[16], own[16];
#pragma unroll
        for (int r = 0; r < 8; r++) { own[r] = m[r] = R[c][r]; own[8 + r] = m[8 + r] = I[c][r]; }
        if (VM) dp_mirror8_valu(own, m); else dp_shfl(m, l ^ 7);

        /* window each block: head with W[i] (block 0: previous shape), tail with W[127-i] */
        const float* ws = tab + AACG_TAB_OFF_WIN_SHORT + 128 * cp[c].shape;
        const float* wh = (w == 0) ? tab + AACG_TAB_OFF_WIN_SHORT + 128 * cp[c].shape_prev : ws;
        float hd[16], tl[16];                         /* [m] = position i = 2g+16m, [8+m] = i+1 */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = 2 * g + 16 * j;
            const dpf2 h0 = *(const dpf2*)(wh + i), h1 = *(const dpf2*)(wh + i + 64);
            const dpf2 t0 = *(const dpf2*)(ws + 126 - i), t1 = *(const dpf2*)(ws + 62 - i);
            hd[j]         = I[c][j + 4] * h0.x;       /* y[2k]       */
            hd[8 + j]     = -m[3 - j] * h0.y;         /* y[2k+1]     */
            hd[j + 4]     = R[c][j] * h1.x;           /* y[64+2k]    */
            hd[8 + j + 4] = -m[8 + 7 - j] * h1.y;     /* y[64+2k+1]  */
            tl[j]         = R[c][j + 4] * t0.y;       /* y[128+2k]   * W[127-i] */
            tl[8 + j]     = -m[8 + 3 - j] * t0.x;     /* y[128+2k+1] * W[126-i] */
            tl[j + 4]     = -I[c][j] * t1.y;          /* y[192+2k]   * W[63-i]  */
            tl[8 + j + 4] = m[7 - j] * t1.x;          /* y[192+2k+1] * W[62-i]  */
        }
        /* s[128 w + i] = tail of block w-1 + head of block w (filter_bank.js:155-160) */
        float pt[16];
#pragma unroll
        for (int i = 0; i < 16; i++) pt[i] = tl[i];
        dp_shfl(pt, (l - 8) & 63);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            hx[c][i] = (w == 0 ? 0.0f : pt[i]) + hd[i];
            hy[c][i] = (w == 0 ? 0.0f : pt[8 + i]) + hd[8 + i];
        }

        /* second half of s -> new overlap (filter_bank.js:164-176) */
        float* tail = area[c];
#pragma unroll
        for (int mm = 0; mm < 8; mm++) {
            const int p = 128 * w + 2 * g + 16 * mm;
            if (p >= 576) { dpf2 t; t.x = hx[c][mm]; t.y = hy[c][mm]; *(dpf2*)(tail + p - 576) = t; }
            if (w == 7)   { dpf2 t; t.x = tl[mm]; t.y = tl[8 + mm]; *(dpf2*)(tail + 448 + 2 * g + 16 * mm) = t; }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {
            const int n = 576 + 2 * l + 128 * t4;
            if (n < 1024) { dpf2 zz; zz.x = 0.0f; zz.y = 0.0f; *(dpf2*)(tail + n) = zz; }
        }
    }
}

/* staging swizzle of the pair planes (defined with the staging helpers below) */
DP_DEVICE int stg(int k);

/* ------------------------------------------------------------------------------------ */
/* CPE with both channels on the same lane map: (left, right) packed arithmetic            */
/* ------------------------------------------------------------------------------------ */
/* Every quantity is a dpv2 = (left, right); one v_pk_*_f32 per operation, no shuffles, and the
 * rotation / twiddle / window factors are shared.  LDS elements are 16 bytes (reL, reR, imL, imR),
 * so a transpose takes half the LDS instructions of two planar ones.
 *
 * Slot layout (2048 floats): the staged spectra as two planes of (L,R) pairs, E[k] = X[2k] at
 * pair index k and O[k] = X[2k+1] at 512 + k; then the transposes (512 x 16 B); then the
 * windowed tails interleaved, pair index n = (tailL[n], tailR[n]). */
struct cpx2 { dpv2 re, im; };

DP_DEVICE dpv2 v2(float a, float b) { dpv2 r; r[0] = a; r[1] = b; return r; }
DP_DEVICE dpv2 v2s(float s) { dpv2 r; r[0] = s; r[1] = s; return r; }
DP_DEVICE void wfix2(dpv2& r, float w) { r = (w == 0.0f) ? v2s(0.0f) : r; }
DP_DEVICE cpx2 c2_add(cpx2 a, cpx2 b) { cpx2 r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
DP_DEVICE cpx2 c2_sub(cpx2 a, cpx2 b) { cpx2 r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
DP_DEVICE cpx2 c2_muli(cpx2 a) { cpx2 r; r.re = -a.im; r.im = a.re; return r; }
DP_DEVICE cpx2 c2_mul(cpx2 a, cpx w)
{
    cpx2 r;
    const dpv2 wr = v2s(w.re), wi = v2s(w.im);
    r.re = a.re * wr - a.im * wi;
    r.im = a.re * wi + a.im * wr;
    return r;
}

DP_DEVICE void radix8_inv2(cpx2 (&x)[8])
{
    const dpv2 h = v2s(0.70710678118654752440f);
    cpx2 a0 = c2_add(x[0], x[4]), a1 = c2_sub(x[0], x[4]);
    cpx2 a2 = c2_add(x[2], x[6]), a3 = c2_sub(x[2], x[6]);
    cpx2 a4 = c2_add(x[1], x[5]), a5 = c2_sub(x[1], x[5]);
    cpx2 a6 = c2_add(x[3], x[7]), a7 = c2_sub(x[3], x[7]);
    cpx2 e0 = c2_add(a0, a2), e2 = c2_sub(a0, a2);
    cpx2 ia3 = c2_muli(a3), ia7 = c2_muli(a7);
    cpx2 e1 = c2_add(a1, ia3), e3 = c2_sub(a1, ia3);
    cpx2 o0 = c2_add(a4, a6), o2 = c2_sub(a4, a6);
    cpx2 o1 = c2_add(a5, ia7), o3 = c2_sub(a5, ia7);
    cpx2 t1, t2, t3;
    t1.re = (o1.re - o1.im) * h;  t1.im = (o1.re + o1.im) * h;
    t2 = c2_muli(o2);
    t3.re = -(o3.re + o3.im) * h; t3.im = (o3.re - o3.im) * h;
    x[0] = c2_add(e0, o0); x[4] = c2_sub(e0, o0);
    x[1] = c2_add(e1, t1); x[5] = c2_sub(e1, t1);
    x[2] = c2_add(e2, t2); x[6] = c2_sub(e2, t2);
    x[3] = c2_add(e3, t3); x[7] = c2_sub(e3, t3);
}

/* 16-byte transposes: slot indices chosen so that ds_write_b128 (8-lane groups, 8 slots) and
 * ds_read_b128 (the four 16-lane groups of MI355X_MICROARCH.md §LDS, 16 slots) are conflict-free */
DP_DEVICE int pch1(int q, int l)     { return 64 * q + (l ^ (((q >> 1) & 1) << 3)); }
DP_DEVICE int pch2(int row, int col) { return 8 * row + (col ^ ((row >> 1) & 7)); }
DP_DEVICE void lds_put2(float* base, int slot, cpx2 v)
{
    dpf4 t; t.x = v.re[0]; t.y = v.re[1]; t.z = v.im[0]; t.w = v.im[1];
    *(dpf4*)(base + 4 * slot) = t;
}
DP_DEVICE cpx2 lds_get2(const float* base, int slot)
{
    const dpf4 t = *(const dpf4*)(base + 4 * slot);
    cpx2 v; v.re = v2(t.x, t.y); v.im = v2(t.z, t.w); return v;
}
DP_DEVICE dpv2 lds_pair(const float* base, int pair_index)
{
    const dpf2 t = *(const dpf2*)(base + 2 * pair_index);
    return v2(t.x, t.y);
}

/* The last FFT stage of the long path leaves lane l with column k = l + 64 i of the spectrum, and the reorder needs the
 * column 63 - l beside it: a whole-wave mirror, 32 ds_bpermute_b32 per channel pair on the CU's one LDS pipe.  Nothing
 * forces column = lane there: with the columns dealt out so that c and 63 - c sit in one row of sixteen lanes at
 * mirrored positions, the exchange is DPP row_mirror on the VALU.  long_col(l): row r = l / 16, position p:
 *   p < 8:  c = 8 r + (p ^ 4 (r & 1));    p >= 8:  c = 63 - long_col(l ^ 15)
 * (the ^ 4 in odd rows keeps every 16-lane read group of ds_read_b128 on 16 distinct 16-byte bank groups: its eight
 * mirror pairs then cover all residues mod 16).  Everything after that stage indexes by column: rotation factors,
 * window positions, tail positions, and the epilogue's sample positions (its `lcol` argument). */
DP_DEVICE int long_col(int l)
{
    const int r = l >> 4, p = l & 15, q = p < 8 ? p : 15 - p;
    const int base = 8 * r + (q ^ ((r & 1) << 2));
    return p < 8 ? base : 63 - base;
}

/* mirror-lane exchange of 8 (L,R) complex values: m*[r] <- lane `src`'s value */
template <int VALU = 0>                                 /* 8 / 16: src is l ^ 7 / l ^ 15, taken without the LDS pipe (DPP) */
DP_DEVICE void shfl_pairs(const dpv2 (&R)[8], const dpv2 (&I)[8], int src, dpv2 (&mR)[8], dpv2 (&mI)[8])
{
    /* the two halves of a pair next to each other: a block of eight moves then fills four register pairs */
    float a[16], b[16], ma[16], mb[16];
#pragma unroll
    for (int r = 0; r < 8; r++) { ma[2 * r] = a[2 * r] = R[r][0]; ma[2 * r + 1] = a[2 * r + 1] = R[r][1]; mb[2 * r] = b[2 * r] = I[r][0]; mb[2 * r + 1] = b[2 * r + 1] = I[r][1]; }
    if (VALU == 8)       { dp_mirror8_valu(a, ma); dp_mirror8_valu(b, mb); }
    else if (VALU == 16) { dp_mirror16_valu(a, ma); dp_mirror16_valu(b, mb); }
    else                 { dp_shfl(ma, src); dp_shfl(mb, src); }
#pragma unroll
    for (int r = 0; r < 8; r++) { mR[r] = v2(ma[2 * r], ma[2 * r + 1]); mI[r] = v2(mb[2 * r], mb[2 * r + 1]); }
}

/* window + reorder of the long pair path for one window sequence (mdct.js:90-114 fused with
 * filter_bank.js:109-141,180-202); tails interleaved into the slot */
template <int SEQ>
DP_DEVICE void long_pair_window(const float* tab, const chan_par& cp, bool want_head, float* slot, int l /* column */,
                                const dpv2 (&R)[8], const dpv2 (&I)[8], const dpv2 (&mR)[8], const dpv2 (&mI)[8],
                                dpv2 (&hx)[8], dpv2 (&hy)[8])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = 2 * l + 128 * j;
        const dpf2 w0 = head_window(tab, SEQ, cp.shape_prev, n), w1 = head_window(tab, SEQ, cp.shape_prev, n + 512);
        const dpf2 v0 = tail_window(tab, SEQ, cp.shape, n), v1 = tail_window(tab, SEQ, cp.shape, n + 512);
        if (want_head) {
            hx[j]     = I[j + 4] * v2s(w0.x);
            hy[j]     = -mR[3 - j] * v2s(w0.y);
            hx[j + 4] = R[j] * v2s(w1.x);
            hy[j + 4] = -mI[7 - j] * v2s(w1.y);
            if (SEQ == AACG_LONG_STOP_SEQUENCE) { wfix2(hx[j], w0.x); wfix2(hy[j], w0.y); wfix2(hx[j + 4], w1.x); wfix2(hy[j + 4], w1.y); }
        }
        dpv2 t0 = R[j + 4] * v2s(v0.x), t1 = -mI[3 - j] * v2s(v0.y);
        dpv2 t2 = -I[j] * v2s(v1.x),    t3 = mR[7 - j] * v2s(v1.y);
        if (SEQ == AACG_LONG_START_SEQUENCE) { wfix2(t0, v0.x); wfix2(t1, v0.y); wfix2(t2, v1.x); wfix2(t3, v1.y); }
        dpf4 o;
        o.x = t0[0]; o.y = t0[1]; o.z = t1[0]; o.w = t1[1];
        *(dpf4*)(slot + 2 * n) = o;                                   /* (tailL[n], tailR[n], tailL[n+1], tailR[n+1]) */
        o.x = t2[0]; o.y = t2[1]; o.z = t3[0]; o.w = t3[1];
        *(dpf4*)(slot + 2 * (n + 512)) = o;
    }
}

/* Long windows, both channels (they share sequence and shapes: one ICSInfo, cpe.js:44, or equal by value). */
template <bool VM>                                      /* VM: columns dealt out by long_col, mirror exchange as DPP row_mirror */
DP_DEVICE void long_pair(const float* tab, const chan_par& cp, bool want_head, float* slot,
                         dpv2 (&hx)[8], dpv2 (&hy)[8])
{
    const int l = dp_lane();
    const float* sincos = tab + AACG_TAB_OFF_SINCOS_LONG;

    /* Slot indices spelled out as a few lane-dependent bases plus compile-time offsets (which become the instructions'
     * immediate offsets): left to itself the compiler rebuilt every swizzled index from the lane number, some 100 vector
     * instructions per frame for index arithmetic alone. */
    cpx2 z[8];
    {
        /* stg(l + 64 j) = 64 j + (l ^ 8 (j & 3));  stg(511 - l - 64 j) = 64 (7 - j) + ((63 - l) ^ 8 (3 - (j & 3))) */
        int eb[4], ob[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { eb[m] = l ^ (m << 3); ob[m] = (63 - l) ^ ((3 - m) << 3); }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const cpx sc = lds_get(sincos, 64 * j + l);
            const dpv2 xe = lds_pair(slot, 64 * j + eb[j & 3]);                       /* X[2k],        k = l + 64 j */
            const dpv2 xo = lds_pair(slot, 512 + 64 * (7 - j) + ob[j & 3]);           /* X[1023 - 2k]               */
            z[j].im = xe * v2s(sc.re) + xo * v2s(sc.im);             /* mdct.js:74 */
            z[j].re = xo * v2s(sc.re) - xe * v2s(sc.im);             /* mdct.js:75 */
        }
    }
    dp_wave_sync();

    radix8_inv2(z);
#pragma unroll
    for (int q = 1; q < 8; q++) z[q] = c2_mul(z[q], lds_get(tab + AACG_TAB_OFF_TW512, 64 * (q - 1) + l));
    {
        const int w0 = l, w1 = l ^ 8;                  /* pch1(q, l) = 64 q + (l ^ 8 (q >> 1 & 1)) */
#pragma unroll
        for (int q = 0; q < 8; q++) lds_put2(slot, 64 * q + ((q & 2) ? w1 : w0), z[q]);
    }
    dp_wave_sync();
    const int l0 = l & 7, qq = l >> 3;
    {
        /* pch1(qq, l0 + 8 j) = 64 qq + l0 + 8 (j ^ (qq >> 1 & 1)): j even + 8, j odd - 8 where that bit is set */
        const int q1 = (qq >> 1) & 1, r0 = 64 * qq + l0 + 8 * q1, r1 = 64 * qq + l0 - 8 * q1;
#pragma unroll
        for (int j = 0; j < 8; j++) z[j] = lds_get2(slot, ((j & 1) ? r1 : r0) + 8 * j);
    }
    dp_wave_sync();
    radix8_inv2(z);
#pragma unroll
    for (int r = 1; r < 8; r++) z[r] = c2_mul(z[r], lds_get(tab + AACG_TAB_OFF_TW64, 8 * (r - 1) + l0));
    {
        /* pch2(qq + 8 r, l0) = 64 r + 8 qq + (l0 ^ (qq >> 1) ^ 4 (r & 1)) */
        const int e0 = 8 * qq + (l0 ^ (qq >> 1)), e1 = e0 ^ 4;
#pragma unroll
        for (int r = 0; r < 8; r++) lds_put2(slot, 64 * r + ((r & 1) ? e1 : e0), z[r]);
    }
    dp_wave_sync();
    const int c = VM ? long_col(l) : l;                /* this lane's column from here on */
    {
        /* pch2(c, i) = (8 c + (c >> 1 & 7)) ^ i, and with the slot 128-byte aligned the byte address is base ^ 16 i */
        const int ab = dp_lds_addr(slot) + 16 * (8 * c + ((c >> 1) & 7));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const dpf4 t = dp_lds_read_f4(ab ^ (16 * i));
            z[i].re = v2(t.x, t.y); z[i].im = v2(t.z, t.w);
        }
    }
    dp_wave_sync();
    radix8_inv2(z);

    dpv2 R[8], I[8], mR[8], mI[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cpx sc = lds_get(sincos, 64 * r + c);
        R[r] = z[r].re * v2s(sc.re) - z[r].im * v2s(sc.im);          /* mdct.js:82-87 */
        I[r] = z[r].im * v2s(sc.re) + z[r].re * v2s(sc.im);
    }
    if (cp.seq == AACG_ONLY_LONG_SEQUENCE) {
        /* the common case as straight-line code: with no branch between them the window reads of an iteration
         * are issued together instead of one LDS round trip each (filter_bank.js:109-116).  Reading all sixteen
         * ahead of the first tail store was measured too: no faster, and 4 more VGPRs. */
        dp_keep_branch();
        const float* wh = tab + AACG_TAB_OFF_WIN_LONG + 1024 * cp.shape_prev;
        const float* wt = tab + AACG_TAB_OFF_WIN_LONG + 1024 * cp.shape;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = 2 * c + 128 * j;
            const dpf2 w0 = *(const dpf2*)(wh + n), w1 = *(const dpf2*)(wh + n + 512);
            const dpf2 r0 = *(const dpf2*)(wt + 1022 - n), r1 = *(const dpf2*)(wt + 510 - n);   /* reversed: (w[1022-n], w[1023-n]) */
            hx[j]     = I[j + 4] * v2s(w0.x);
            hx[j + 4] = R[j] * v2s(w1.x);
            const dpv2 t0 = R[j + 4] * v2s(r0.y), t2 = -I[j] * v2s(r1.y);
            dpv2 t1, t3;
            if (VM) {
                /* the four values of the column 63 - c: fetched from the mirror lane and windowed in one instruction each */
                const float src[8] = {R[3 - j][0], R[3 - j][1], I[7 - j][0], I[7 - j][1], I[3 - j][0], I[3 - j][1], R[7 - j][0], R[7 - j][1]};
                const float wv[4] = {w0.y, w1.y, r0.x, r1.x};
                float o[8];
                dp_window_mirror<16>(src, wv, o, j == 0);
                hy[j] = v2(o[0], o[1]); hy[j + 4] = v2(o[2], o[3]); t1 = v2(o[4], o[5]); t3 = v2(o[6], o[7]);
            } else {
                if (j == 0) shfl_pairs<0>(R, I, 63 - l, mR, mI);
                hy[j]     = -mR[3 - j] * v2s(w0.y);
                hy[j + 4] = -mI[7 - j] * v2s(w1.y);
                t1 = -mI[3 - j] * v2s(r0.x); t3 = mR[7 - j] * v2s(r1.x);
            }
            dpf4 o;
            o.x = t0[0]; o.y = t0[1]; o.z = t1[0]; o.w = t1[1];
            *(dpf4*)(slot + 2 * n) = o;
            o.x = t2[0]; o.y = t2[1]; o.z = t3[0]; o.w = t3[1];
            *(dpf4*)(slot + 2 * (n + 512)) = o;
        }
        return;
    }
    shfl_pairs<VM ? 16 : 0>(R, I, 63 - l, mR, mI);
    /* LONG_START / LONG_STOP: the same loop with the sequence as a compile-time constant, so that the branches
     * inside head_window / tail_window fold and each iteration's window reads are issued together */
    if (cp.seq == AACG_LONG_START_SEQUENCE) long_pair_window<AACG_LONG_START_SEQUENCE>(tab, cp, want_head, slot, c, R, I, mR, mI, hx, hy);
    else                                    long_pair_window<AACG_LONG_STOP_SEQUENCE>(tab, cp, want_head, slot, c, R, I, mR, mI, hx, hy);
}

/* EIGHT_SHORT, both channels.  Slot indices as a few lane-dependent bases plus compile-time offsets, like long_pair: with
 * stg / pch2 evaluated per element the path carried some 150 vector instructions of index arithmetic per frame. */
template <bool VM>
DP_DEVICE void short_pair(const float* tab, const chan_par& cp, float* slot, dpv2 (&hx)[8], dpv2 (&hy)[8])
{
    const int l = dp_lane(), w = l >> 3, g = l & 7;
    const float* sincos = tab + AACG_TAB_OFF_SINCOS_SHORT;

    cpx2 z[8];
    {
        /* stg(64 w + g + 8 j)      = 64 w + g + 8 (j ^ (w & 3)):             base eb[j & 3] + 32 (j >> 2)
         * stg(64 w + 63 - g - 8 j) = 64 w + 7 - g + 8 ((7 - j) ^ (w & 3)):   base ob[3 - (j & 3)] + 32 (1 - (j >> 2)) */
        const int w3 = (w & 3) << 3, e0 = ((w << 6) | g) | w3, o0 = (512 + ((w << 6) | (7 - g))) | w3;
        int eb[4], ob[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { eb[m] = e0 ^ (m << 3); ob[m] = o0 ^ (m << 3); }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const cpx sc = lds_get(sincos, 8 * j + g);
            const dpv2 xe = lds_pair(slot, eb[j & 3] + 32 * (j >> 2));                 /* X_w[2k], k = g + 8 j */
            const dpv2 xo = lds_pair(slot, ob[3 - (j & 3)] + 32 * (1 - (j >> 2)));     /* X_w[127 - 2k]        */
            z[j].im = xe * v2s(sc.re) + xo * v2s(sc.im);
            z[j].re = xo * v2s(sc.re) - xe * v2s(sc.im);
        }
    }
    dp_wave_sync();

    radix8_inv2(z);
#pragma unroll
    for (int q = 1; q < 8; q++) z[q] = c2_mul(z[q], lds_get(tab + AACG_TAB_OFF_TW64, 8 * (q - 1) + g));
    {
        /* pch2(8 w + q, g) = 64 w + 8 q + (g ^ 4 (w & 1) ^ (q >> 1)): base pb[q >> 1] + 8 q */
        const int p0 = (w << 6) | (g ^ ((w & 1) << 2));
        int pb[4];
#pragma unroll
        for (int m = 0; m < 4; m++) pb[m] = p0 ^ m;
#pragma unroll
        for (int q = 0; q < 8; q++) lds_put2(slot, pb[q >> 1] + 8 * q, z[q]);
    }
    dp_wave_sync();
    {
        /* pch2(l, i) = (8 l + (l >> 1 & 7)) ^ i; with the slot 128-byte aligned the byte address is base ^ 16 i */
        const int ab = dp_lds_addr(slot) + 16 * (8 * l + ((l >> 1) & 7));
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const dpf4 t = dp_lds_read_f4(ab ^ (16 * i));
            z[i].re = v2(t.x, t.y); z[i].im = v2(t.z, t.w);
        }
    }
    dp_wave_sync();
    radix8_inv2(z);

    dpv2 R[8], I[8], mR[8], mI[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cpx sc = lds_get(sincos, 8 * r + g);
        R[r] = z[r].re * v2s(sc.re) - z[r].im * v2s(sc.im);
        I[r] = z[r].im * v2s(sc.re) + z[r].re * v2s(sc.im);
    }
    if (!VM) shfl_pairs<0>(R, I, l ^ 7, mR, mI);

    const float* ws = tab + AACG_TAB_OFF_WIN_SHORT + 128 * cp.shape;
    const float* wh = (w == 0) ? tab + AACG_TAB_OFF_WIN_SHORT + 128 * cp.shape_prev : ws;
    dpv2 hd[16], tl[16];                               /* [m] = position i = 2g+16m, [8+m] = i+1 */
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = 2 * g + 16 * j;
        const dpf2 h0 = *(const dpf2*)(wh + i), h1 = *(const dpf2*)(wh + i + 64);
        const dpf2 t0 = *(const dpf2*)(ws + 126 - i), t1 = *(const dpf2*)(ws + 62 - i);
        hd[j]         = I[j + 4] * v2s(h0.x);
        hd[j + 4]     = R[j] * v2s(h1.x);
        tl[j]         = R[j + 4] * v2s(t0.y);
        tl[j + 4]     = -I[j] * v2s(t1.y);
        if (VM) {
            /* the values of lane ^ 7 (the window's column 63 - k), fetched and windowed in one instruction each */
            const float src[8] = {R[3 - j][0], R[3 - j][1], I[7 - j][0], I[7 - j][1], I[3 - j][0], I[3 - j][1], R[7 - j][0], R[7 - j][1]};
            const float wv[4] = {h0.y, h1.y, t0.x, t1.x};
            float o[8];
            dp_window_mirror<8>(src, wv, o, j == 0);
            hd[8 + j] = v2(o[0], o[1]); hd[8 + j + 4] = v2(o[2], o[3]); tl[8 + j] = v2(o[4], o[5]); tl[8 + j + 4] = v2(o[6], o[7]);
        } else {
            hd[8 + j]     = -mR[3 - j] * v2s(h0.y);
            hd[8 + j + 4] = -mI[7 - j] * v2s(h1.y);
            tl[8 + j]     = -mI[3 - j] * v2s(t0.x);
            tl[8 + j + 4] = mR[7 - j] * v2s(t1.x);
        }
    }
    /* s[128 w + i] = tail of block w-1 + head of block w (filter_bank.js:155-160): the previous window's tails sit eight lanes
     * down.  The rotation goes through the wave's own slot, which is free between the last transpose read and the tails:
     * eight 16-byte stores ([m][lane]: conflict-free) + eight loads instead of 32 ds_bpermute (a rotation by eight lanes
     * crosses the 16-lane rows, so it has no DPP form).  The first window has no block before it and adds
     * zeros: on the int16 seam (VM) its lanes skip the loads (one lane mask around them) instead of selecting 32 registers
     * afterwards; the f32 kernels keep the selects — with the mask their register allocation went from 123 to 128 with spills
     * in the _dd and _rv variants (tests/test_kernel_resources.py). */
    dpv2 pt[16];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        dpf4 o; o.x = tl[2 * m][0]; o.y = tl[2 * m][1]; o.z = tl[2 * m + 1][0]; o.w = tl[2 * m + 1][1];
        *(dpf4*)(slot + 256 * m + 4 * l) = o;
    }
    dp_wave_sync();
    if (VM) {
#pragma unroll
        for (int m = 0; m < 16; m++) pt[m] = v2s(0.0f);
        if (w != 0) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const dpf4 t = *(const dpf4*)(slot + 256 * m + 4 * (l - 8));
                pt[2 * m] = v2(t.x, t.y); pt[2 * m + 1] = v2(t.z, t.w);
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const dpf4 t = *(const dpf4*)(slot + 256 * m + 4 * ((l - 8) & 63));
            pt[2 * m] = v2(t.x, t.y); pt[2 * m + 1] = v2(t.z, t.w);
        }
    }
    dp_wave_sync();                                    /* the slot takes the tails next */
#pragma unroll
    for (int i = 0; i < 8; i++) {
        hx[i] = ((!VM && w == 0) ? v2s(0.0f) : pt[i]) + hd[i];
        hy[i] = ((!VM && w == 0) ? v2s(0.0f) : pt[8 + i]) + hd[8 + i];
    }

    /* second half of s -> new overlap, interleaved (filter_bank.js:164-176): position p = 128 w + 2 g + 16 mm takes part
     * from 576 on, i.e. windows 5.. with every mm and window 4 with mm >= 4 — two lane masks instead of eight compares */
    if (w >= 4) {
#pragma unroll
        for (int mm = 4; mm < 8; mm++) {
            const int p = 128 * w + 2 * g + 16 * mm;
            dpf4 o; o.x = hx[mm][0]; o.y = hx[mm][1]; o.z = hy[mm][0]; o.w = hy[mm][1]; *(dpf4*)(slot + 2 * (p - 576)) = o;
        }
    }
    if (w >= 5) {
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            const int p = 128 * w + 2 * g + 16 * mm;
            dpf4 o; o.x = hx[mm][0]; o.y = hx[mm][1]; o.z = hy[mm][0]; o.w = hy[mm][1]; *(dpf4*)(slot + 2 * (p - 576)) = o;
        }
    }
    if (w == 7) {
#pragma unroll
        for (int mm = 0; mm < 8; mm++) { dpf4 o; o.x = tl[mm][0]; o.y = tl[mm][1]; o.z = tl[8 + mm][0]; o.w = tl[8 + mm][1]; *(dpf4*)(slot + 2 * (448 + 2 * g + 16 * mm)) = o; }
    }
    {
        dpf4 zz; zz.x = zz.y = zz.z = zz.w = 0.0f;
#pragma unroll
        for (int t4 = 0; t4 < 3; t4++) *(dpf4*)(slot + 2 * (576 + 2 * l + 128 * t4)) = zz;        /* 576 .. 959 */
        if (l < 32) *(dpf4*)(slot + 2 * (960 + 2 * l)) = zz;                                      /* 960 .. 1023 */
    }
}

/* ------------------------------------------------------------------------------------ */
/* spectral reconstruction: dequant (ics.js:222-227,244-256), MS (decoder.js:379-404),       */
/* IS (decoder.js:337-376) in natural order: lane l owns coefficients 8 l + 512 i + 0..7     */
/* ------------------------------------------------------------------------------------ */
/* Per-wave band records in LDS (in the work area, before the spectrum is staged): for channel c and band
 * index b = g * maxSFB + sfb (ics.js:217) two words at bt + 256 c + 2 b:
 *   val   = the band's scale (sf, sign applied) if the band carries coefficients; the intensity scale
 *           +-sf (decoder.js:353-368) on an intensity band of the right channel; else 0
 *   flags = bit 31: the band carries coefficients (band type 1..12);  bit 0: left channel: MS applies to
 *           this band (decoder.js:295-296,393); right channel: intensity band
 * Everything that depends only on the band is decided here once (two bands per lane) instead of once per
 * coefficient group.  Record 127 is the "no band" record (sfb >= maxSFB): dead, no MS, no IS. */
#define AACG_BR_LIVE    0x80000000u
#define AACG_BR_FLAG    0x00000001u
#define AACG_BR_NONE    127
struct quant_regs { dpi4 ql[2], qr[2]; unsigned mw[2][2]; };

/* the unit's quantised spectra (16 bytes per lane per load) and raw band words, issued early */
template <bool NTL = false>                             /* NTL: non-temporal loads (dp_load_nt) */
DP_DEVICE void quant_load(const void* coeffs, const aacg_band_meta* metas, uint32_t coef_block, uint32_t meta_block, int n_ch, quant_regs& r)
{
    const int lane = dp_lane();
    const int16_t* q0 = (const int16_t*)coeffs + (size_t)coef_block * 1024u;
    const aacg_band_meta* meta = metas + meta_block;
    /* unconditional loads (a single channel reads its own block twice): no per-load branches, so the
     * compiler keeps all of them in flight together */
    const int16_t* q1 = q0 + (n_ch == 2 ? 1024 : 0);
    const aacg_band_meta* m1 = meta + (n_ch == 2 ? 1 : 0);
    const int b1 = lane + 64 < AACG_MAX_SECTIONS ? lane + 64 : AACG_MAX_SECTIONS - 1;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (NTL) { r.ql[i] = dp_load_nt((const dpi4*)(q0 + 8 * lane + 512 * i)); r.qr[i] = dp_load_nt((const dpi4*)(q1 + 8 * lane + 512 * i)); }
        else {
        r.ql[i] = *(const dpi4*)(q0 + 8 * lane + 512 * i);
        r.qr[i] = *(const dpi4*)(q1 + 8 * lane + 512 * i);
        }
    }
    r.mw[0][0] = meta->band[lane]; r.mw[0][1] = meta->band[b1];
    r.mw[1][0] = m1->band[lane];   r.mw[1][1] = m1->band[b1];
}

struct chan_ctx {
    int cls;            /* 1 = EIGHT_SHORT */
    int max_sfb;
    unsigned gmap;      /* 4 bits per window: its group (planner-filled) */
};

/* The raw band-map bytes a lane needs for its four 4-coefficient groups (k = 2 i + h: positions
 * 8 lane + 512 i + 4 h).  The maps hold one byte per coefficient, so one 8-byte read covers the two groups of
 * an i; both the long and the short map are read unconditionally (no branch between the loads). */
struct band_raw { dpf2 lng[2], sht; };
DP_DEVICE void band_raw_load(const float* tab, band_raw& r)
{
    const int lane = dp_lane();
    const unsigned char* bl = (const unsigned char*)(tab + AACG_TAB_OFF_BAND_LONG);
    const unsigned char* bs = (const unsigned char*)(tab + AACG_TAB_OFF_BAND_SHORT);
    r.lng[0] = *(const dpf2*)(bl + 8 * lane);
    r.lng[1] = *(const dpf2*)(bl + 8 * lane + 512);
    r.sht = *(const dpf2*)(bs + ((8 * lane) & 127));   /* pos & 127 does not depend on i */
}
/* Band record indices g * maxSFB + sfb (ics.js:217); AACG_BR_NONE if sfb >= maxSFB. */
DP_DEVICE void band_indices(const band_raw& r, const chan_ctx& cc, int (&idx)[4])
{
    const int lane = dp_lane();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const dpf2 m = cc.cls ? r.sht : r.lng[i];
        const int s0 = (int)(__builtin_bit_cast(unsigned, m.x) & 0xffu), s1 = (int)(__builtin_bit_cast(unsigned, m.y) & 0xffu);
        /* short: window = pos >> 7 = lane / 16 + 4 i, its group from the map; long: group 0 */
        const int g = cc.cls ? (int)((cc.gmap >> (4 * ((lane >> 4) + 4 * i))) & 15u) : 0;
        idx[2 * i]     = s0 < cc.max_sfb ? g * cc.max_sfb + s0 : AACG_BR_NONE;
        idx[2 * i + 1] = s1 < cc.max_sfb ? g * cc.max_sfb + s1 : AACG_BR_NONE;
    }
}

/* Band records of both channels, two bands per lane; sl / sr: the bands' SF-table entries (positive), loaded by the
 * caller together with the band maps.  Everything is a bit field of the two band words (include/aacgpu.h: scalefactor
 * index, negate, ms_used, band type), so the decisions are shifts of small constants by the band type instead of
 * compare chains — `K >> type & 1` with K the set of types for which the property holds; the wave-uniform conditions
 * (two channels, MS possible, mask present) are folded into those constants on the scalar unit. */
template <int H0, int H1>
DP_DEVICE void prepare_bands(const quant_regs& r, const float* tab, bool two, bool ms_on, bool mask, float* bt)
{
    const int lane = dp_lane();
    unsigned* bw = (unsigned*)bt;
    const unsigned kCoded = 0x1FFEu;                   /* band types 1..12 carry coefficients (ics.js:222-227) */
    const unsigned kBelowNoise = 0x1FFFu;              /* 0..12 */
    const unsigned kLiveR = two ? kCoded : 0u, kMsL = ms_on ? kBelowNoise : 0u;
    const unsigned kIs = two ? 0xC000u : 0u;           /* INTENSITY_BT2, INTENSITY_BT on the right channel */
    const unsigned mask_u = mask ? 1u : 0u;
    /* one batch of independent LDS reads: the SF-table entries of this lane's bands */
    float sl_in[2], sr_in[2];
#pragma unroll
    for (int h = H0; h < H1; h++) {
        sl_in[h] = tab[AACG_TAB_OFF_SF + (r.mw[0][h] & AACG_META_SF_MASK)];
        sr_in[h] = tab[AACG_TAB_OFF_SF + (r.mw[1][h] & AACG_META_SF_MASK)];
    }
#pragma unroll
    for (int h = H0; h < H1; h++) {
        const int b = lane + 64 * h;
        const bool coded = b < AACG_MAX_SECTIONS;      /* the rest (incl. AACG_BR_NONE) are empty records: band type 0 */
        const unsigned wl = coded ? r.mw[0][h] : 0u, wr = coded ? r.mw[1][h] : 0u;
        const unsigned tl = wl >> AACG_META_BT_SHIFT, tr = wr >> AACG_META_BT_SHIFT;
        const unsigned live_l = 0u - ((kCoded >> tl) & 1u), live_r = 0u - ((kLiveR >> tr) & 1u);       /* all ones / zero */
        const unsigned ms_used = (wl / AACG_META_MS_USED) & 1u;
        /* decoder.js:295-296,393: MS needs commonWindow && maskPresent && ms_used[idx] && both band types < NOISE */
        const unsigned ms = ms_used & (kMsL >> tl) & (kBelowNoise >> tr) & 1u;
        /* decoder.js:353-368: right = left * (c * sf) on the right channel's intensity bands; c = -1 for
         * INTENSITY_BT2, flipped again where the mask is present and ms_used is set */
        const unsigned is = (kIs >> tr) & 1u;
        const unsigned neg = ((0x4000u >> tr) ^ (mask_u & ms_used)) & 1u;
        /* the scalefactor with the word's negate bit as its sign (the table holds positive values) */
        const unsigned sl = __builtin_bit_cast(unsigned, sl_in[h]) ^ ((wl * (0x80000000u / AACG_META_NEGATE)) & 0x80000000u);
        const unsigned sr = __builtin_bit_cast(unsigned, sr_in[h]) ^ ((wr * (0x80000000u / AACG_META_NEGATE)) & 0x80000000u);
        dpf2 rl, rr;
        rl.x = __builtin_bit_cast(float, sl & live_l);
        rl.y = __builtin_bit_cast(float, (live_l & AACG_BR_LIVE) | ms);
        rr.x = __builtin_bit_cast(float, (sr & live_r) | ((sr ^ (neg << 31)) & (0u - is)));
        rr.y = __builtin_bit_cast(float, (live_r & AACG_BR_LIVE) | is);
        *(dpf2*)(bw + 2 * b) = rl;
        *(dpf2*)(bw + 256 + 2 * b) = rr;
    }
}

/* sign(q) |q|^(4/3) sf for four coefficients packed as two dwords of int16 pairs, branch-free and without
 * unpacking: v_mad_i32_i16 turns each half straight into the LDS byte address of its entry in the signed
 * table (q = -512..511, `iq0` = address of the q = 0 entry), one fused multiply-add applies the scalefactor.
 *   live band:  x = IQ[q] * sf + (-0)   (q == 0 gives -0 like ics.js:251: the table holds -0 there)
 *   dead band:  x = IQ[q] * 0  + (+0) = +0   (uncoded, ZERO or INTENSITY band, ics.js:222-227)
 * Larger magnitudes read some other LDS word (or 0 outside the allocation); `oor` collects the packed
 * q + 512 so that the caller can detect them afterwards and patch those elements (dequant4_big). */
DP_DEVICE void dequant4(int iq0, float sf_eff, float z_eff, int p01, int p23, float (&x)[4], int& oor)
{
    oor |= dp_pk_add_u16(p01, 0x02000200) | dp_pk_add_u16(p23, 0x02000200);
    dpv2 a, b, sf, z;
    a[0] = dp_lds_read_f32(dp_mad4_i16_lo(p01, iq0)); a[1] = dp_lds_read_f32(dp_mad4_i16_hi(p01, iq0));
    b[0] = dp_lds_read_f32(dp_mad4_i16_lo(p23, iq0)); b[1] = dp_lds_read_f32(dp_mad4_i16_hi(p23, iq0));
    sf[0] = sf[1] = sf_eff; z[0] = z[1] = z_eff;
    a = dp_fma2(a, sf, z);                             /* v_pk_fma_f32: two coefficients per instruction */
    b = dp_fma2(b, sf, z);
    x[0] = a[0]; x[1] = a[1]; x[2] = b[0]; x[3] = b[1];
}
#define AACG_OOR_MASK ((int)0xFC00FC00)                /* some q + 512 outside 0..1023 */

/* the rare magnitudes outside the LDS table: full IQ_TABLE in global memory; [8191] = NaN like the JS
 * out-of-range read */
DP_DEVICE void dequant4_big(const aacg_tables* T, bool live, float sf, int p01, int p23, float (&x)[4])
{
    const int q[4] = {(int)(short)(p01 & 0xffff), p01 >> 16, (int)(short)(p23 & 0xffff), p23 >> 16};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int a = q[e] < 0 ? -q[e] : q[e];
        if ((q[e] + 512) & ~1023) {
            const float v = T->iq[a > 8191 ? 8191 : a];
            x[e] = live ? (q[e] > 0 ? v : -v) * sf : 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* AACG_PNS_SPEC: NOISE_BT bands as ics.js:228-243 was meant to fill them                     */
/* ------------------------------------------------------------------------------------ */
/* exclusive prefix sum over the wave's 128 band slots: two values per lane (slots lane and lane + 64) */
DP_DEVICE void wave_excl_scan2(int& v0, int& v1)
{
    const int lane = dp_lane();
    int in0 = v0, in1 = v1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float t[2] = {__builtin_bit_cast(float, in0), __builtin_bit_cast(float, in1)};
        dp_shfl(t, (lane - d) & 63);
        if (lane >= d) { in0 += __builtin_bit_cast(int, t[0]); in1 += __builtin_bit_cast(int, t[1]); }
    }
    float tot[1] = {__builtin_bit_cast(float, in0)};   /* inclusive sums; lane 63 holds the first half's total */
    dp_shfl(tot, 63);
    v0 = in0 - v0;
    v1 = in1 - v1 + __builtin_bit_cast(int, tot[0]);
}

/* One channel's noise bands, in place on the natural-order registers x (8 lane + 512 i + e).  The reference's
 * generator restarts for every channel of every frame (fresh ICStream, decoder.js:145,153), so draw number p is
 * the table entry rnd[p]; a band-window takes `width` consecutive draws, in the order of ics.js:214-243 (group,
 * band, window of the group, coefficient), and is scaled to sf / sqrt(sum of squares).  nb: LDS scratch, 256 ints. */
DP_DEVICE void pns_channel(const aacg_pns_tables* T, const float* tab, const chan_ctx& cc, const unsigned (&mw)[2],
                           int* nb, float (&x)[16])
{
    const int lane = dp_lane();
    const uint16_t* swb = cc.cls ? T->swb_short : T->swb_long;
    /* windows per group, and each window's rank inside its group, from the 4-bit-per-window group map */
    auto group_len = [&](int g) { int n = 0; for (int w = 0; w < 8; w++) n += (cc.cls && ((cc.gmap >> (4 * w)) & 15u) == (unsigned)g) ? 1 : 0; return cc.cls ? n : 1; };
    /* 1. draws per band slot, exclusive prefix sum -> first draw of every band; kept with the noise scale */
    int cnt[2]; bool noise[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int b = lane + 64 * h;
        const unsigned wd = mw[h];
        noise[h] = b < AACG_MAX_SECTIONS && (int)(wd >> AACG_META_BT_SHIFT) == AACG_NOISE_BT && cc.max_sfb > 0;
        const int g = cc.max_sfb > 0 ? b / cc.max_sfb : 0, sfb = cc.max_sfb > 0 ? b % cc.max_sfb : 0;
        cnt[h] = noise[h] ? group_len(g) * ((int)swb[sfb + 1] - (int)swb[sfb]) : 0;
    }
    int first[2] = {cnt[0], cnt[1]};
    wave_excl_scan2(first[0], first[1]);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int b = lane + 64 * h;
        nb[b] = noise[h] ? first[h] : -1;
        float sf = tab[AACG_TAB_OFF_SF + (mw[h] & AACG_META_SF_MASK)];
        if (mw[h] & AACG_META_NEGATE) sf = -sf;
        ((float*)nb)[128 + b] = sf;
    }
    dp_wave_sync();
    /* 2. this lane's four 4-coefficient groups */
    band_raw braw;
    band_raw_load(tab, braw);
    int idx[4];
    band_indices(braw, cc, idx);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = k >> 1, h = k & 1;
        const int pos = 8 * lane + 512 * i + 4 * h;
        const int p_band = idx[k] == AACG_BR_NONE ? -1 : nb[idx[k]];
        if (p_band >= 0) {
            const int w = cc.cls ? pos >> 7 : 0;
            const int g = cc.cls ? (int)((cc.gmap >> (4 * w)) & 15u) : 0;
            int rank = 0;                               /* windows of the same group before this one */
            for (int v = 0; v < 8; v++) rank += (cc.cls && v < w && ((cc.gmap >> (4 * v)) & 15u) == (unsigned)g) ? 1 : 0;
            const int sfb = idx[k] - g * cc.max_sfb;
            const int width = (int)swb[sfb + 1] - (int)swb[sfb];
            const int pw = p_band + rank * width;       /* first draw of this band-window */
            const int k0 = (cc.cls ? (pos & 127) : pos) - (int)swb[sfb];
            const double energy = T->esum[pw + width] - T->esum[pw];
            const double scale = (double)((const float*)nb)[128 + idx[k]] / __builtin_sqrt(energy);
#pragma unroll
            for (int e = 0; e < 4; e++) x[4 * k + e] = (float)((double)T->rnd[(pw + k0 + e) & 1023] * scale);
        }
    }
    dp_wave_sync();
}

/* Produces xl / xr[16]: element 8 i + e is coefficient 8 lane + 512 i + e of the left / right
 * (or single) channel after dequant, MS and IS. */
template <bool PNS = false>
DP_DEVICE void spectral_quant(const aacg_kparams& P, const float* tab, const unit_view& u, int n_ch,
                              const quant_regs& qreg, float* bt, float (&xl)[16], float (&xr)[16])
{
    chan_ctx ccL, ccR;
    ccL.cls = u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE; ccL.max_sfb = u.max_sfb[0]; ccL.gmap = u.gmap[0];
    ccR.cls = u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE; ccR.max_sfb = u.max_sfb[1]; ccR.gmap = u.gmap[1];

    const dpi4 (&ql)[2] = qreg.ql;
    const dpi4 (&qr)[2] = qreg.qr;
    const bool two = n_ch == 2;
    const bool ms_on = two && (u.flags & AACG_UNIT_COMMON_WINDOW) && (u.flags & AACG_UNIT_MASK_PRESENT);
    const bool mask  = (u.flags & AACG_UNIT_MASK_PRESENT) != 0;
    band_raw braw;
    band_raw_load(tab, braw);
    /* band indices g * maxSFB + sfb reach 64 and beyond only with window groups (8 x 15 sections); a long window has
     * at most 51 bands, so its frames prepare one band per lane and the "no band" record (wave-uniform) */
    if (ccL.cls | ccR.cls) {
        dp_keep_branch();
        prepare_bands<0, 2>(qreg, tab, two, ms_on, mask, bt);
    } else {
        prepare_bands<0, 1>(qreg, tab, two, ms_on, mask, bt);
        dpf2 none; none.x = none.y = 0.0f;
        *(dpf2*)(bt + 2 * (64 + dp_lane())) = none;
        *(dpf2*)(bt + 256 + 2 * (64 + dp_lane())) = none;
    }
    int idxL[4], idxR[4];
    band_indices(braw, ccL, idxL);
#pragma unroll
    for (int k = 0; k < 4; k++) idxR[k] = idxL[k];
    if (ccR.cls != ccL.cls || ccR.max_sfb != ccL.max_sfb || ccR.gmap != ccL.gmap) {   /* wave-uniform; a common window never gets here */
        dp_keep_branch();
        band_indices(braw, ccR, idxR);
    }
    dp_wave_sync();

    /* per 4-coefficient group (bands are multiples of 4 wide): the band records of both channels, all loads
     * unconditional and independent of each other */
    dpf2 recL[4], recR[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        recL[k] = *(const dpf2*)(bt + 2 * idxL[k]);
        recR[k] = *(const dpf2*)(bt + 256 + 2 * idxR[k]);
    }
    dp_lanes g_ms[4], g_is[4];
    bool  liveL[4], liveR[4], isR[4];
    float g_isc[4], sfL[4], sfR[4];
    int big = 0;
    const int iq0 = dp_lds_addr(tab + AACG_TAB_OFF_IQ_SMALL + 512);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = k >> 1, h = k & 1;
        const unsigned fl = __builtin_bit_cast(unsigned, recL[k].y), fr = __builtin_bit_cast(unsigned, recR[k].y);
        g_ms[k] = dp_lanes_where((fl & AACG_BR_FLAG) != 0);
        g_is[k] = dp_lanes_where((fr & AACG_BR_FLAG) != 0);
        isR[k] = (fr & AACG_BR_FLAG) != 0;
        liveL[k] = (fl & AACG_BR_LIVE) != 0;
        liveR[k] = (fr & AACG_BR_LIVE) != 0;
        sfL[k] = recL[k].x;
        g_isc[k] = recR[k].x;
        sfR[k] = isR[k] ? 0.0f : recR[k].x;
        /* the addend is -0 on a band that carries coefficients, +0 elsewhere: the flag's sign bit */
        dequant4(iq0, sfL[k], __builtin_bit_cast(float, fl & AACG_BR_LIVE), h ? ql[i].z : ql[i].x, h ? ql[i].w : ql[i].y, *(float (*)[4])&xl[4 * k], big);
        dequant4(iq0, sfR[k], __builtin_bit_cast(float, fr & AACG_BR_LIVE), h ? qr[i].z : qr[i].x, h ? qr[i].w : qr[i].y, *(float (*)[4])&xr[4 * k], big);
    }

    if (dp_any((big & AACG_OOR_MASK) != 0)) {          /* escape-coded magnitudes: rare */
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = k >> 1, h = k & 1;
            dequant4_big(P.tab, liveL[k], sfL[k], h ? ql[i].z : ql[i].x, h ? ql[i].w : ql[i].y, *(float (*)[4])&xl[4 * k]);
            dequant4_big(P.tab, liveR[k], sfR[k], h ? qr[i].z : qr[i].x, h ? qr[i].w : qr[i].y, *(float (*)[4])&xr[4 * k]);
        }
    }

    if (PNS && (u.flags & AACG_UNIT_HAS_PNS)) {
        /* the noise bands come before MS / IS like everything decodeSpectralData produces: an intensity band of
         * the right channel copies whatever the left channel holds there (decoder.js:353-368), noise included.
         * bt + 512: scratch behind the band records (the PNS kernel gives every wave 1024 floats) */
        pns_channel(P.pns, tab, ccL, qreg.mw[0], (int*)(bt + 512), xl);
        if (two) pns_channel(P.pns, tab, ccR, qreg.mw[1], (int*)(bt + 512), xr);
    }
    if (two) {
        /* wave-uniform skips: most frames carry no intensity bands, many no MS */
        if (dp_lanes_any(g_ms[0] | g_ms[1] | g_ms[2] | g_ms[3])) {
#pragma unroll
            for (int k = 0; k < 4; k++) {              /* sums and differences two at a time, under the group's lane mask */
                dpv2 a0 = v2(xl[4 * k], xl[4 * k + 1]), a1 = v2(xl[4 * k + 2], xl[4 * k + 3]);
                dpv2 b0 = v2(xr[4 * k], xr[4 * k + 1]), b1 = v2(xr[4 * k + 2], xr[4 * k + 3]);
                dp_sumdiff_where(g_ms[k], a0, a1, b0, b1);
                xl[4 * k] = a0[0]; xl[4 * k + 1] = a0[1]; xl[4 * k + 2] = a1[0]; xl[4 * k + 3] = a1[1];
                xr[4 * k] = b0[0]; xr[4 * k + 1] = b0[1]; xr[4 * k + 2] = b1[0]; xr[4 * k + 3] = b1[1];
            }
        }
        if (dp_lanes_any(g_is[0] | g_is[1] | g_is[2] | g_is[3])) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const dpv2 l0 = v2(xl[4 * k], xl[4 * k + 1]), l1 = v2(xl[4 * k + 2], xl[4 * k + 3]);
                dpv2 r0 = v2(xr[4 * k], xr[4 * k + 1]), r1 = v2(xr[4 * k + 2], xr[4 * k + 3]);
                dp_scale_where(g_is[k], l0, l1, g_isc[k], r0, r1);
                xr[4 * k] = r0[0]; xr[4 * k + 1] = r0[1]; xr[4 * k + 2] = r1[0]; xr[4 * k + 3] = r1[1];
            }
        }
    }
    dp_wave_sync();                                    /* band table dead: the area may be overwritten */
}

/* ------------------------------------------------------------------------------------ */
/* staging: spectra -> LDS                                                                 */
/* ------------------------------------------------------------------------------------ */
/* pair planes: pair index k of X[2k] (E plane, at 0) / X[2k+1] (O plane, at 512); the XOR keeps both the
 * long (l + 64 j) and the short (64 w + g + 8 j) read patterns conflict-free */
DP_DEVICE int stg(int k) { return k ^ (((k >> 6) & 3) << 3); }

/* registers in natural order (8 lane + 512 i + e) -> planar area[0..1023] */
DP_DEVICE void stage_nat8(const float (&x)[16], float* area)
{
    const int lane = dp_lane();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        dpf4 a, b;
        a.x = x[8 * i]; a.y = x[8 * i + 1]; a.z = x[8 * i + 2]; a.w = x[8 * i + 3];
        b.x = x[8 * i + 4]; b.y = x[8 * i + 5]; b.z = x[8 * i + 6]; b.w = x[8 * i + 7];
        *(dpf4*)(area + 8 * lane + 512 * i) = a;
        *(dpf4*)(area + 8 * lane + 512 * i + 4) = b;
    }
}
/* the same for a CPE on the pair path: (L,R) pairs into the E / O planes */
DP_DEVICE void stage_pair_nat8(const float (&xl)[16], const float (&xr)[16], float* slot)
{
    const int lane = dp_lane();
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 4 * lane + 256 * i + 2 * h, e = 8 * i + 4 * h;
            dpf4 ev, od;
            ev.x = xl[e];     ev.y = xr[e];     ev.z = xl[e + 2]; ev.w = xr[e + 2];
            od.x = xl[e + 1]; od.y = xr[e + 1]; od.z = xl[e + 3]; od.w = xr[e + 3];
            *(dpf4*)(slot + 2 * stg(k)) = ev;
            *(dpf4*)(slot + 2 * (512 + stg(k))) = od;
        }
    }
}
/* f32 spectra as loaded (float4 at 4 lane + 256 i per channel) */
DP_DEVICE void stage_pair_f32(const dpf4 (&xa)[4], const dpf4 (&xb)[4], float* slot)
{
    const int lane = dp_lane();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k = 2 * lane + 128 * i;
        dpf4 ev, od;
        ev.x = xa[i].x; ev.y = xb[i].x; ev.z = xa[i].z; ev.w = xb[i].z;
        od.x = xa[i].y; od.y = xb[i].y; od.z = xa[i].w; od.w = xb[i].w;
        *(dpf4*)(slot + 2 * stg(k)) = ev;
        *(dpf4*)(slot + 2 * (512 + stg(k))) = od;
    }
}

/* ------------------------------------------------------------------------------------ */
/* AACG_TNS_SPEC: the all-pole filter tns.js:155-163 was meant to run                        */
/* ------------------------------------------------------------------------------------ */
/* y[m] = x[m] - sum_{k=1..min(m,order)} lpc[k-1] y[m-k] over `size` samples in processing order (position =
 * start + inc * m): a serial recurrence per (window, filter).  Both channels of a unit are filtered side by side, one
 * per half of the wave (lanes 0..31 and 32..63), on their spectra in ICStream.data order in the wave's slot.
 *
 * Long windows (up to three filters per channel, one pass each): lane b of a half takes block b = 32 consecutive
 * samples.
 *   1. zero-state response of the block (float32, tap by tap like tns.js:160); its last P outputs are c_b;
 *   2. the block transition is the same matrix M = A^32 for every block (A = companion matrix of the filter): lane
 *      r < P computes ROW r of it in double precision by 32 steps of the row recurrence w <- w A (in float32 the
 *      transitions of a near-unstable filter lose the state: 2e-3 of the signal against 2e-6);
 *   3. the true block-end states by a serial carry v_b = M v_(b-1) + c_b: per step the P row lanes read v_(b-1)
 *      (P floats, a broadcast read from a small exchange buffer), add their element of c_b and write v_b; lane b + 1
 *      picks v_b up as its incoming state.  P^2 multiply-adds per block — the block scan this replaces (Hillis-Steele over 64
 *      blocks of 16 with the matrix squared between levels) spent log2(64) P^2 per block on every lane and read the
 *      matrices from LDS for each: 97 us per config-3 batch, bound by the LDS broadcast reads;
 *   4. the block again from its true incoming state, same arithmetic as step 1, written back in place.
 * EIGHT_SHORT (one filter per window, 128 samples, order <= 7): lane w of a half runs window w serially in place,
 * four samples per trip — no transition matrices at all.
 * xch: the wave's exchange area, AACG_SPX_XCH_FLOATS floats (per half: the c_b of a round of blocks). */
#define AACG_TNS_BLOCK 32
#define AACG_TNS_XCH_FLOATS(R) (2 * (R) * AACG_TNS_MAX_ORDER)   /* per wave: two halves of [R][P] floats (the c_b of a round of blocks) */

/* chunk c (four samples in processing order) of the run that starts at blk */
DP_DEVICE void tns_chunk_load(const float* blk, int inc, int c, float (&x)[4])
{
    const dpf4 t = *(const dpf4*)(blk + 4 * c * inc);
    if (inc > 0) { x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
    else         { x[0] = t.w; x[1] = t.z; x[2] = t.y; x[3] = t.x; }
}
DP_DEVICE void tns_chunk_store(float* blk, int inc, int c, const float (&y)[4])
{
    dpf4 t;
    if (inc > 0) { t.x = y[0]; t.y = y[1]; t.z = y[2]; t.w = y[3]; }
    else         { t.w = y[0]; t.z = y[1]; t.y = y[2]; t.x = y[3]; }
    *(dpf4*)(blk + 4 * c * inc) = t;
}

/* `chunks` chunks of four samples of the recurrence, starting from the state h (h[k] = y[-1-k]) and leaving the state
 * behind the last chunk in h; STORE: outputs written back in place.  Chunks at or beyond n_valid samples take zeros in
 * and store nothing.  float32, tap by tap like tns.js:160 (a zero state contributes exact zeros). */
template <int P, bool STORE>
DP_DEVICE void tns_run(float* blk, int inc, int chunks, int n_valid, const float (&lpc)[P], float (&h)[P])
{
#pragma unroll 1
    for (int c = 0; c < chunks; c++) {
        float x[4] = {0.0f, 0.0f, 0.0f, 0.0f}, y[4];
        if (4 * c < n_valid) tns_chunk_load(blk, inc, c, x);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float acc = x[j];
#pragma unroll
            for (int k = 1; k <= P; k++) acc = dp_fma(-lpc[k - 1], (k <= j) ? y[j - k] : h[k - j - 1], acc);
            y[j] = acc;
        }
        if (STORE && 4 * c < n_valid) tns_chunk_store(blk, inc, c, y);
#pragma unroll
        for (int k = P - 1; k >= 4; k--) h[k] = h[k - 4];
        h[0] = y[3]; h[1] = y[2]; h[2] = y[1]; h[3] = y[0];
    }
}

/* Row r of M = A^BL, the transition of a block of BL samples (A: the filter's companion matrix — first row -lpc, ones below
 * the diagonal): w <- w A, BL times, from e_r, in double precision (in float32 the transitions of a near-unstable filter lose
 * the state: 2e-3 of the signal against 2e-6).  It depends on the filter only, not on the data: 768 of a wave's ~7000 issue
 * slots per pass when every wave derived it for itself, so it is made ONCE per plan by a kernel of its own
 * (aacg_tns_matrices, aacg_engine_spectral.hip: one lane per row, this very function) and read back by the row lanes —
 * 96 bytes per lane, requested before the block's zero-state pass and in the registers after it. */
#define AACG_TNS_M_DOUBLES (3 * AACG_TNS_MAX_ORDER * AACG_TNS_MAX_ORDER)      /* per channel record: [filter slot 0..2][row][column] */
/* where they ride in the kernel arguments of the launches that run filters (aacg_set_tns_m, aacg_device.h) */
DP_DEVICE const double* aacg_tns_m(const aacg_kparams& P) { return (const double*)(const void*)P.scratch; }
DP_DEVICE void tns_matrix_row(const float (&lpc)[AACG_TNS_MAX_ORDER], int r, double (&row)[AACG_TNS_MAX_ORDER])
{
    constexpr int P = AACG_TNS_MAX_ORDER, BL = 32;
#pragma unroll
    for (int k = 0; k < P; k++) row[k] = (k == r) ? 1.0 : 0.0;
#pragma unroll 2
    for (int step = 0; step < BL; step++) {
        const double w0 = row[0];
#pragma unroll
        for (int k = 0; k < P - 1; k++) row[k] = dp_fma(-w0, (double)lpc[k], row[k + 1]);
        row[P - 1] = -w0 * (double)lpc[P - 1];
    }
}
/* the matrices of one channel record's long-window filter slots: lane = 16 f + r of the record's 64 */
DP_DEVICE void tns_matrices_body(const aacg_dev_tns* recs, double* M, uint32_t n_records)
{
    constexpr int P = AACG_TNS_MAX_ORDER;
    const uint32_t rec = (uint32_t)dp_block() * AACG_WG_WAVES + (uint32_t)dp_wave();
    const int lane = dp_lane(), f = lane >> 4, r = lane & 15;
    if (rec >= n_records || f >= 3 || r >= P) return;
    const aacg_dev_tns* t = recs + rec;
    const int order = t->order[f];
    float lpc[P];
#pragma unroll
    for (int k = 0; k < P; k++) lpc[k] = (k < order) ? t->lpc[f][k] : 0.0f;
    double row[P];
    tns_matrix_row(lpc, r, row);
    double* dst = M + (size_t)rec * AACG_TNS_M_DOUBLES + (size_t)(f * P + r) * P;
#pragma unroll
    for (int k = 0; k < P; k++) dst[k] = row[k];
}

/* one long-window filter of each channel (slot f of the records; order 0 / null record = nothing to do in that half).
 * mA / mB: the records' transition matrices (AACG_TNS_M_DOUBLES each; null where the record is) */
template <int R>
DP_DEVICE void tns_long_pass(float* slot, float* xch, const aacg_dev_tns* recA, const aacg_dev_tns* recB, int f, const double* mA, const double* mB)
{
    constexpr int P = AACG_TNS_MAX_ORDER, BL = AACG_TNS_BLOCK;
    const int lane = dp_lane(), half = lane >> 5, b = lane & 31;
    /* per-half parameters: wave-uniform loads, selected by half */
    const int orderA = recA ? recA->order[f] : 0, orderB = recB ? recB->order[f] : 0;
    if (orderA <= 0 && orderB <= 0) return;
    const int sizeA = orderA > 0 ? recA->size[f] : 0, sizeB = orderB > 0 ? recB->size[f] : 0;
    const int n_steps = (((sizeA > sizeB ? sizeA : sizeB) + BL - 1) / BL) - 1;       /* carry steps: blocks of the longer filter - 1 */
    const aacg_dev_tns* rec = half ? recB : recA;
    const int order = half ? orderB : orderA, size = half ? sizeB : sizeA;
    const int start = order > 0 ? rec->start[f] : 0, inc = order > 0 ? rec->inc[f] : 1;
    float lpc[P];
#pragma unroll
    for (int k = 0; k < P; k++) lpc[k] = (k < order) ? rec->lpc[f][k] : 0.0f;
    float* area = slot + 1024 * half;
    float* cbuf = xch + AACG_TNS_XCH_FLOATS(R) / 2 * half;   /* c_b of a round's blocks, [R][P] floats */

    /* this lane's block, in processing order; blocks before the last are full */
    const int m0 = BL * b;
    const int n_valid = order > 0 ? (size - m0 < 0 ? 0 : (size - m0 > BL ? BL : size - m0)) : 0;
    float* blk = area + (inc > 0 ? start + m0 : start - m0 - 3);

    /* 2 (its loads first). row r of M = A^BL (tns_matrix_row).  Every row of sixteen lanes holds the twelve rows (lane r of
     * the row: row r): the carry below then never leaves a lane row.  Lanes r16 >= P and halves without a filter: zeros,
     * they ride along. */
    const int r16 = lane & 15;
    double row[P];
    const double* mrec = half ? mB : mA;
    {
        const bool have = mrec != nullptr && order > 0 && r16 < P;
        const double* src = have ? mrec + (size_t)(f * P + r16) * P : (mA ? mA : mB);     /* (always a readable address: one of the two is a record's) */
#pragma unroll
        for (int k = 0; k < P; k += 2) { const dpd2 t = *(const dpd2*)(src + k); row[k] = have ? t.x : 0.0; row[k + 1] = have ? t.y : 0.0; }
    }

    /* 1. zero-state response; the state it leaves is the block's own contribution c_b to the state behind it */
    float c_own[P];
#pragma unroll
    for (int k = 0; k < P; k++) c_own[k] = 0.0f;
    tns_run<P, false>(blk, inc, BL / 4, n_valid, lpc, c_own);


    /* 3. serial carry: v_0 = c_0; v_s = M v_(s-1) + c_s.  s_in = this block's incoming state (y[-1-k]).
     * What this loop costs is LDS time, one pipe for the CU's 16 waves (a wave-wide 16-byte store takes 13 cycles to
     * issue whatever its exec mask, a 16-byte read 4): so c_s goes to the row lanes transposed once per ROUND of
     * R blocks (lane b of the round stores its c_own, row lane r picks element r of each), the state
     * travels as 12 floats (one 4-byte store per step by the row lanes, three 16-byte reads by everybody: the row
     * lanes take it as M's operand, lane s keeps it as its incoming state), and M stays in double in the row lanes'
     * registers.  23 LDS cycles per step against 95 when lane s published c_s and the state went round as doubles
     * (config-3 batch with a filter everywhere: 40.0 -> 36.1 us).  M and its products in float32 were measured too:
     * nothing on encoder-like filters, but a filter drawn from the full coefficient table left the state 1e-2 off
     * (1e-4 in double), and the serial recurrence of steps 1 and 4 in the other tap order: no faster. */
    float s_in[P];
#pragma unroll
    for (int k = 0; k < P; k++) s_in[k] = 0.0f;
    float vmine = 0.0f;                                  /* lane r of a lane row: v_(s-1)[r] */
#pragma unroll 1
    for (int g = 0; R * g <= n_steps; g++) {
        if (b / R == g) {
#pragma unroll
            for (int k = 0; k < P; k += 4) { dpf4 t; t.x = c_own[k]; t.y = c_own[k + 1]; t.z = c_own[k + 2]; t.w = c_own[k + 3]; *(dpf4*)(cbuf + P * (b % R) + k) = t; }
        }
        dp_wave_sync();
        float cs[R];
#pragma unroll
        for (int j = 0; j < R; j++) cs[j] = cbuf[P * j + (r16 < P ? r16 : 0)];
        dp_wave_sync();
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int st = R * g + j;
            if (st <= n_steps) {                         /* wave-uniform */
                double acc0 = (double)cs[j], acc1 = 0.0, acc2 = 0.0;
                if (st > 0) {
                    /* v_(s-1) from the twelve row lanes of this lane's own row of sixteen: DPP broadcasts on the VALU
                     * (round 2 kept the state in LDS: a store, a wave-wide sync and three loads on every step's critical
                     * path — the carry is one long dependency chain, its latency is what the filter costs) */
                    float f[P];
                    dp_row_gather12(vmine, f);
#pragma unroll
                    for (int k = 0; k < P; k++) s_in[k] = (b == st) ? f[k] : s_in[k];
#pragma unroll
                    for (int t = 0; t < P; t += 3) {     /* three short chains instead of one of P */
                        acc0 = dp_fma(row[t], (double)f[t], acc0);
                        acc1 = dp_fma(row[t + 1], (double)f[t + 1], acc1);
                        acc2 = dp_fma(row[t + 2], (double)f[t + 2], acc2);
                    }
                }
                vmine = (float)(acc0 + (acc1 + acc2));
            }
        }
    }

    /* 4. the block from its true incoming state, same arithmetic as step 1 (read again rather than held in registers
     * across the carry), written back in place */
    tns_run<P, true>(blk, inc, BL / 4, n_valid, lpc, s_in);
    dp_wave_sync();
}

/* EIGHT_SHORT channels (shortA / shortB): lane w of the half filters window w (record slot w) serially, in place */
DP_DEVICE void tns_short_pass(float* slot, const aacg_dev_tns* recA, const aacg_dev_tns* recB, bool shortA, bool shortB)
{
    constexpr int P = 8;                               /* AAC-LC's short-window limit is 7 */
    const int lane = dp_lane(), half = lane >> 5, w = lane & 31;
    const aacg_dev_tns* rec = half ? recB : recA;
    const bool mine = (half ? shortB : shortA) && rec != nullptr && w < 8;
    const int order = mine ? rec->order[w] : 0;
    const int size = order > 0 ? rec->size[w] : 0;
    const int start = order > 0 ? rec->start[w] : 0, inc = order > 0 ? rec->inc[w] : 1;
    float lpc[P];
#pragma unroll
    for (int k = 0; k < P; k++) lpc[k] = (k < order) ? rec->lpc[w][k] : 0.0f;
    float* blk = slot + 1024 * half + (inc > 0 ? start : start - 3);
    float h[P];                                        /* h[k] = y[m - 1 - k] */
#pragma unroll
    for (int k = 0; k < P; k++) h[k] = 0.0f;
    if (dp_any(size > 0)) tns_run<P, true>(blk, inc, 32, size, lpc, h);      /* a window is 128 samples: at most 32 chunks */
    dp_wave_sync();
}

/* All TNS filters of a unit, in place on its spectra in the slot (channel c at slot + 1024 c, ICStream.data order).
 * rec0 / rec1: the channels' records, null where a channel has none. */
template <int R>
DP_DEVICE void tns_unit(float* slot, float* xch, const aacg_dev_tns* rec0, const aacg_dev_tns* rec1, bool short0, bool short1,
                        const aacg_dev_tns* recs = nullptr, const double* M = nullptr)
{
    const aacg_dev_tns* l0 = short0 ? nullptr : rec0;
    const aacg_dev_tns* l1 = short1 ? nullptr : rec1;
    /* the records' transition matrices: same index as the record */
    const double* m0 = (M && l0) ? M + (size_t)(l0 - recs) * AACG_TNS_M_DOUBLES : nullptr;
    const double* m1 = (M && l1) ? M + (size_t)(l1 - recs) * AACG_TNS_M_DOUBLES : nullptr;
    if (l0 || l1)
        for (int f = 0; f < 3; f++) tns_long_pass<R>(slot, xch, l0, l1, f, m0, m1);   /* up to three filters, disjoint band ranges (tns.js:119-124) */
    if ((short0 && rec0) || (short1 && rec1)) tns_short_pass(slot, rec0, rec1, short0, short1);
}

#define AACG_RUN_TNS_ROUND  3

/* IMDCT + window of a unit whose spectra are staged in its slot.  CPE tails always end up
 * interleaved (pair index n = (tailL[n], tailR[n])); a single channel's tail is planar. */
#ifdef AACG_VM_F32
#define AACG_VM_KIND(kind) true
#else
#define AACG_VM_KIND(kind) ((kind) == AACG_INPUT_QUANT_I16)
#endif
/* VM: the mirror exchanges of the short windows on the VALU: worth it where the LDS pipe is the busier one — the int16 seam
 * (all-short batch 14.67 -> 13.70 us, config 3 13.94 -> 13.56); the f32 seam lost 0.07 us with it and keeps ds_bpermute */
template <bool VM>
DP_DEVICE void filter_unit(const float* tab, const unit_view& u, int n_ch, bool pair_path, bool want_head, float* slot,
                           float (&hx0)[8], float (&hy0)[8], float (&hx1)[8], float (&hy1)[8])
{
    chan_par p0, p1;
    p0.seq = u.seq[0]; p0.shape = u.shape[0]; p0.shape_prev = u.shape_prev[0];
    p1.seq = u.seq[1]; p1.shape = u.shape[1]; p1.shape_prev = u.shape_prev[1];
    const bool s0 = p0.seq == AACG_EIGHT_SHORT_SEQUENCE, s1 = p1.seq == AACG_EIGHT_SHORT_SEQUENCE;
    if (pair_path) {
        dpv2 hx[8], hy[8];
        if (s0) short_pair<VM>(tab, p0, slot, hx, hy);
        else    long_pair<VM>(tab, p0, want_head, slot, hx, hy);
#pragma unroll
        for (int m = 0; m < 8; m++) { hx0[m] = hx[m][0]; hx1[m] = hx[m][1]; hy0[m] = hy[m][0]; hy1[m] = hy[m][1]; }
        return;
    }
    {
        const chan_par cp[1] = {p0};
        float* const area[1] = {slot};
        float hx[1][8], hy[1][8];
        if (s0) short_channels<1, VM>(tab, cp, area, hx, hy);
        else    long_channels<1, VM>(tab, cp, want_head, area, hx, hy);
#pragma unroll
        for (int m = 0; m < 8; m++) { hx0[m] = hx[0][m]; hy0[m] = hy[0][m]; }
    }
    if (n_ch == 2) {
        const chan_par cp[1] = {p1};
        float* const area[1] = {slot + 1024};
        float hx[1][8], hy[1][8];
        if (s1) short_channels<1, VM>(tab, cp, area, hx, hy);
        else    long_channels<1, VM>(tab, cp, want_head, area, hx, hy);
#pragma unroll
        for (int m = 0; m < 8; m++) { hx1[m] = hx[0][m]; hy1[m] = hy[0][m]; }
        /* two planar tails -> the interleaved form the next wave expects */
        const int lane = dp_lane();
        dpf4 a[4], b[4];
        dp_wave_sync();
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = *(const dpf4*)(slot + 4 * lane + 256 * i); b[i] = *(const dpf4*)(slot + 1024 + 4 * lane + 256 * i); }
        dp_wave_sync();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dpf4 o;
            o.x = a[i].x; o.y = b[i].x; o.z = a[i].y; o.w = b[i].y;
            *(dpf4*)(slot + 2 * (4 * lane + 256 * i)) = o;
            o.x = a[i].z; o.y = b[i].z; o.z = a[i].w; o.w = b[i].w;
            *(dpf4*)(slot + 2 * (4 * lane + 256 * i) + 4) = o;
        }
    }
}

/* Overlap-add in place: this unit's windowed first half onto the previous frame's tails in the slot of the wave that made them
 * (filter_bank.js:109-111, 153-160, 185-195 — both operands are PCM-scaled already), which then holds the element's
 * finished samples: (L[n], R[n]) pairs for a CPE, planar for a single channel.  Samples a window sequence takes from the
 * overlap alone (EIGHT_SHORT: 0..447) are simply left as they are. */
DP_DEVICE void overlap_add_in_place(float* prev, int n_ch, int cls0, int cls1, int lcol,
                              const float (&hx0)[8], const float (&hy0)[8], const float (&hx1)[8], const float (&hy1)[8])
{
    /* the overlap-add is an addition of two rounded floats (filter_bank.js:109-111 adds a stored product): never a
     * multiply-add fused with the window product, whatever the compiler finds next to it — the fused and the staged
     * coupling routes, the first and the later frames of a chain must produce the same bits */
#pragma clang fp contract(off)
    const int lane = dp_lane(), w = lane >> 3, g = lane & 7;
    (void)lane;
    if (n_ch == 2 && cls0 == cls1) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int n = cls0 ? 448 + 128 * w + 2 * g + 16 * m : 2 * lcol + 128 * m;
            if (!cls0 || w < 4 || (w == 4 && m < 4)) {
                dpf4 v = *(const dpf4*)(prev + 2 * n);
                v.x += hx0[m]; v.y += hx1[m]; v.z += hy0[m]; v.w += hy1[m];
                *(dpf4*)(prev + 2 * n) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c < n_ch) {
            const int cls = c ? cls1 : cls0;
            const float (&hx)[8] = c ? hx1 : hx0;
            const float (&hy)[8] = c ? hy1 : hy0;
            float* p = prev + c;                       /* single channel: planar; CPE with mixed lane maps: stride 2 */
            const int st = n_ch;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int n = cls ? 448 + 128 * w + 2 * g + 16 * m : 2 * lcol + 128 * m;
                if (!cls || w < 4 || (w == 4 && m < 4)) {
                    p[n * st] += hx[m];
                    p[(n + 1) * st] += hy[m];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* epilogue: out = (overlap + head) / 32768, interleaved (filter_bank.js + decoder.js:203-215) */
/* ------------------------------------------------------------------------------------ */
/* (ov0[n], ov1[n], ov0[n+1], ov1[n+1]) of the incoming tails: FROM_LDS: the previous wave's slot
 * (interleaved for a CPE, planar for one channel); else the overlap state in HBM (planar). */
template <bool FROM_LDS, bool SC1 = false>            /* SC1: a rendezvous payload another workgroup wrote in this launch: agent-scope loads */
DP_DEVICE dpf4 incoming(const float* p0, const float* p1, int n_ch, int n)
{
    dpf4 r;
    if (FROM_LDS && n_ch == 2) return *(const dpf4*)(p0 + 2 * n);
    const dpf2 a = (!FROM_LDS && SC1) ? dp_g_load_f2(p0 + n) : *(const dpf2*)(p0 + n);
    r.x = a.x; r.z = a.y; r.y = 0.0f; r.w = 0.0f;
    if (n_ch == 2) { const dpf2 b = (!FROM_LDS && SC1) ? dp_g_load_f2(p1 + n) : *(const dpf2*)(p1 + n); r.y = b.x; r.w = b.y; }
    return r;
}

/* PCM goes out as float32 (the reference's Float32Array, decoder.js:204) or, for engines created with
 * AACG_OUTPUT_I16, as int16 (round to nearest, saturating): the same sample positions, narrower elements. */
template <int OUT> struct pcm_elem { typedef float type; };
template <> struct pcm_elem<AACG_OUTPUT_I16> { typedef int16_t type; };
DP_DEVICE void pcm_put4(float* p, float a, float b, float c, float d) { dpf4 o; o.x = a; o.y = b; o.z = c; o.w = d; dp_store_nt((dpf4*)p, o); }   /* four adjacent samples, 16-byte aligned position (8 for int16) */
DP_DEVICE void pcm_put4(int16_t* p, float a, float b, float c, float d) { dp_store_i2_nt(p, dp_pcm16_pair(a, b), dp_pcm16_pair(c, d)); }
DP_DEVICE void pcm_put2(float* p, float a, float b) { dp_store2_u(p, a, b); }                      /* two adjacent samples */
DP_DEVICE void pcm_put2(int16_t* p, float a, float b) { dp_store_i1_u(p, dp_pcm16_pair(a, b)); }
DP_DEVICE void pcm_put1(float* p, float a) { *p = a; }
DP_DEVICE void pcm_put1(int16_t* p, float a) { *p = (int16_t)(dp_pcm16_pair(a, a) & 0xffff); }

/* AACG_CCE_SPEC, independent coupling where the target's PCM is formed (CPL builds of the run kernels): a sample of a target
 * channel takes data[n] += gain * cce.data[n] (cce.js:121-128; on a Float32Array: one fused multiply-add) for each of its
 * unit's jobs, in the order of the frame's coupling elements, on the finished sample (overlap + windowed first half),
 * PCM-scaled like the side buffer.  The job list is wave-uniform: scalar loads. */

/* the first two coupling jobs of a unit, their coupling element's samples lane + 64 j requested before the wave waits for its
 * predecessor's tails (a job's loads are a memory round trip: inside the epilogue they were 37 us of a 7-channel batch) */
struct cpl_prefetch { float sv[2][16]; float gain[2]; int second[2]; int n; };
template <bool CPL>
DP_DEVICE void couple_prefetch(const aacg_kparams& P, const unit_view& u, cpl_prefetch& pre)
{
    pre.n = 0;
    if (!CPL) return;
    const int lane = dp_lane();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if ((uint32_t)k < u.cpl_n) {
            const aacg_couple_job* job = AACG_CPL_JOBS(P) + (u.cpl_first + k);
            const float* src = AACG_CPL_SIDE(P) + (size_t)job->src * 1024u;
            pre.gain[k] = AACG_CPL_GAINS(P)[job->gain_off];
            pre.second[k] = job->dst != 0;
#pragma unroll
            for (int j = 0; j < 16; j++) pre.sv[k][j] = src[lane + 64 * j];
            pre.n = k + 1;
        }
    }
}

template <bool FROM_LDS, int OUT, bool CPL = false, bool SC1 = false>
DP_DEVICE void epilogue(const aacg_kparams& P, const cpl_prefetch& pre, const float* p0, const float* p1, const unit_view& u, int n_ch, int cls0, int cls1,
                        float* pcm_base_f32, const float (&hx0)[8], const float (&hy0)[8],
                        const float (&hx1)[8], const float (&hy1)[8], int lcol /* the lane's column in the long lane map: long_col(lane) or lane */)
{
    /* overlap + windowed first half: an addition of two rounded floats, never fused with the window product (see
     * overlap_add_in_place); the coupling terms are explicit fused multiply-adds (dp_fma), which the pragma leaves alone */
#pragma clang fp contract(off)
    /* a unit with coupling jobs: the paths that finish a sample in one place — in place in the previous wave's slot, or the
     * per-channel scalar path of a chain's first frame — so that the jobs are applied to finished samples, in order */
    const bool coupled = CPL && u.cpl_n > 0;
    typedef typename pcm_elem<OUT>::type elem;
    elem* pcm_base = (elem*)pcm_base_f32;
    const int lane = dp_lane(), w = lane >> 3, g = lane & 7;
    /* decoder.js:211's 1 / 32768 is already in the windows (AACG_PCM_SCALE): heads and tails arrive PCM-scaled */
    const int C = u.n_out_ch;
    elem* pcm = pcm_base + u.pcm_offset + u.channel;

    if (n_ch == 2 && C == 2 && !coupled && cls0 == cls1 && ((u.pcm_offset | (uint32_t)u.channel) & 3u) == 0) {
        /* stereo fast path: (L[n], R[n], L[n+1], R[n+1]) = 16 bytes per lane */
        if (!cls0) {
            /* all eight reads of the incoming tails first: one LDS (or HBM) round trip, not one per store */
            dpf4 v[8];
#pragma unroll
            for (int m = 0; m < 8; m++) v[m] = incoming<FROM_LDS, SC1>(p0, p1, 2, 2 * lcol + 128 * m);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int n = 2 * lcol + 128 * m;
                pcm_put4(pcm + 2 * n, v[m].x + hx0[m], v[m].y + hx1[m], v[m].z + hy0[m], v[m].w + hy1[m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (w < 4 || (w == 4 && m < 4)) {
                    const int n = 448 + 128 * w + 2 * g + 16 * m;
                    const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, 2, n);
                    pcm_put4(pcm + 2 * n, v.x + hx0[m], v.y + hx1[m], v.z + hy0[m], v.w + hy1[m]);
                }
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {           /* out[0..447] = overlap (filter_bank.js:149-151) */
                const int n = 2 * lane + 128 * t4;
                if (n < 448) {
                    const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, 2, n);
                    pcm_put4(pcm + 2 * n, v.x, v.y, v.z, v.w);
                }
            }
        }
        return;
    }

    if (FROM_LDS) {
        /* An element of a wider frame (5.1 etc.) whose incoming tails sit in the previous wave's slot, which nobody
         * reads after this wave: the overlap-add happens in place there, and the finished samples are read back with
         * CONSECUTIVE samples in consecutive lanes.  A store instruction then covers 64 x 4 C contiguous bytes (14 lines
         * for C = 7) instead of the IMDCT lane map's two samples per lane at 8 C bytes from lane to lane (28 lines): the
         * multichannel layouts are bound by the L2's write requests, not by bytes (14.8 M requests per config-5 launch,
         * 7.9 M this way: 93 -> 66.5 us).  Non-temporal stores here: 150 us (they bypass the L2's merging of the
         * elements' partial lines). */
        float* prev = const_cast<float*>(p0);
        overlap_add_in_place(prev, n_ch, cls0, cls1, lcol, hx0, hy0, hx1, hy1);
        dp_wave_sync();
        if (coupled) {
            /* the unit's finished samples in registers (sample lane + 64 j of channel 0 / 1), then job after job: its gain
             * and side block as scalars, sixteen coalesced loads of the coupling element's samples, sixteen multiply-adds */
            float a[16], b[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n = lane + 64 * j;
                if (n_ch == 2) { const dpf2 lr = *(const dpf2*)(prev + 2 * n); a[j] = lr.x; b[j] = lr.y; }
                else           { a[j] = prev[n]; b[j] = 0.0f; }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {                  /* the prefetched jobs (couple_prefetch) */
                if (k < pre.n) {
#pragma unroll
                    for (int j = 0; j < 16; j++) { if (pre.second[k]) b[j] = dp_fma(pre.gain[k], pre.sv[k][j], b[j]); else a[j] = dp_fma(pre.gain[k], pre.sv[k][j], a[j]); }
                }
            }
            for (uint32_t k = (uint32_t)pre.n; k < u.cpl_n; k++) {
                const aacg_couple_job* job = AACG_CPL_JOBS(P) + (u.cpl_first + k);
                const float gain = AACG_CPL_GAINS(P)[job->gain_off];
                const float* src = AACG_CPL_SIDE(P) + (size_t)job->src * 1024u;
                const bool second = job->dst != 0;
                float sv[16];
#pragma unroll
                for (int j = 0; j < 16; j++) sv[j] = src[lane + 64 * j];
#pragma unroll
                for (int j = 0; j < 16; j++) { if (second) b[j] = dp_fma(gain, sv[j], b[j]); else a[j] = dp_fma(gain, sv[j], a[j]); }
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n = lane + 64 * j;
                if (n_ch == 2) pcm_put2(pcm + (size_t)n * C, a[j], b[j]); else pcm_put1(pcm + (size_t)n * C, a[j]);
            }
            return;
        }
        if (n_ch == 2) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n = lane + 64 * j;
                const dpf2 lr = *(const dpf2*)(prev + 2 * n);
                pcm_put2(pcm + (size_t)n * C, lr.x, lr.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int n = lane + 64 * j;
                pcm_put1(pcm + (size_t)n * C, prev[n]);
            }
        }
        return;
    }

    if (n_ch == 2 && cls0 == cls1 && !coupled) {
        /* a CPE inside a wider frame (5.1 etc.): (L[n], R[n]) are adjacent, one 8-byte store per sample
         * (4-byte aligned when the channel count is odd) */
        if (!cls0) {
            /* all reads of the incoming tails first: the stores may alias them as far as the compiler knows, and a read
             * behind every store is a memory round trip each (a chain's first frame takes its state from HBM) */
            dpf4 v[8];
#pragma unroll
            for (int m = 0; m < 8; m++) v[m] = incoming<FROM_LDS, SC1>(p0, p1, 2, 2 * lcol + 128 * m);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int n = 2 * lcol + 128 * m;
                pcm_put2(pcm + (size_t)n * C, v[m].x + hx0[m], v[m].y + hx1[m]);
                pcm_put2(pcm + (size_t)(n + 1) * C, v[m].z + hy0[m], v[m].w + hy1[m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                if (w < 4 || (w == 4 && m < 4)) {
                    const int n = 448 + 128 * w + 2 * g + 16 * m;
                    const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, 2, n);
                    pcm_put2(pcm + (size_t)n * C, v.x + hx0[m], v.y + hx1[m]);
                    pcm_put2(pcm + (size_t)(n + 1) * C, v.z + hy0[m], v.w + hy1[m]);
                }
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; t4++) {
                const int n = 2 * lane + 128 * t4;
                if (n < 448) {
                    const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, 2, n);
                    pcm_put2(pcm + (size_t)n * C, v.x, v.y);
                    pcm_put2(pcm + (size_t)(n + 1) * C, v.z, v.w);
                }
            }
        }
        return;
    }

    if (coupled) {
        /* the first frame of a chain with coupling jobs (overlap from the state buffer; FROM_LDS units took the in-place
         * path above): per channel the lane's finished samples (n, n + 1) at up to twelve positions — eight of the
         * windowed first half, and for EIGHT_SHORT four more of the stretch the overlap alone fills (filter_bank.js:149-151)
         * — then job after job (gain and side block as scalars, the loads issued together), then the stores */
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c < n_ch) {
                elem* dst = pcm + c;
                const int cls = c ? cls1 : cls0;
                const float (&hx)[8] = c ? hx1 : hx0;
                const float (&hy)[8] = c ? hy1 : hy0;
                int pos[12]; bool ok[12]; float x[12], y[12];
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    const int m = i & 7, t4 = i - 8;
                    if (i < 8) { pos[i] = cls ? 448 + 128 * w + 2 * g + 16 * m : 2 * lcol + 128 * m; ok[i] = !cls || w < 4 || (w == 4 && m < 4); }
                    else       { pos[i] = 2 * lane + 128 * t4; ok[i] = cls && pos[i] < 448; }
                    if (!ok[i]) pos[i] = 0;
                    const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, n_ch, pos[i]);
                    x[i] = (c ? v.y : v.x) + (i < 8 ? hx[m] : 0.0f);
                    y[i] = (c ? v.w : v.z) + (i < 8 ? hy[m] : 0.0f);
                }
                for (uint32_t k = 0; k < u.cpl_n; k++) {
                    const aacg_couple_job* job = AACG_CPL_JOBS(P) + (u.cpl_first + k);
                    if ((int)job->dst != c) continue;
                    const float gain = AACG_CPL_GAINS(P)[job->gain_off];
                    const float* src = AACG_CPL_SIDE(P) + (size_t)job->src * 1024u;
                    dpf2 sv[12];
#pragma unroll
                    for (int i = 0; i < 12; i++) sv[i] = *(const dpf2*)(src + pos[i]);
#pragma unroll
                    for (int i = 0; i < 12; i++) { x[i] = dp_fma(gain, sv[i].x, x[i]); y[i] = dp_fma(gain, sv[i].y, y[i]); }
                }
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    if (ok[i]) { pcm_put1(dst + (size_t)pos[i] * C, x[i]); pcm_put1(dst + (size_t)(pos[i] + 1) * C, y[i]); }
                }
            }
        }
        return;
    }

    /* single channels / mixed lane maps: scalar stores at stride C */
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c < n_ch) {
            elem* dst = pcm + c;
            const int cls = c ? cls1 : cls0;
            const float (&hx)[8] = c ? hx1 : hx0;
            const float (&hy)[8] = c ? hy1 : hy0;
            if (!cls) {
                dpf4 v[8];                              /* reads first, as above */
#pragma unroll
                for (int m = 0; m < 8; m++) v[m] = incoming<FROM_LDS, SC1>(p0, p1, n_ch, 2 * lcol + 128 * m);
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int n = 2 * lcol + 128 * m;
                    pcm_put1(dst + (size_t)n * C, (c ? v[m].y : v[m].x) + hx[m]);
                    pcm_put1(dst + (size_t)(n + 1) * C, (c ? v[m].w : v[m].z) + hy[m]);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    if (w < 4 || (w == 4 && m < 4)) {
                        const int n = 448 + 128 * w + 2 * g + 16 * m;
                        const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, n_ch, n);
                        pcm_put1(dst + (size_t)n * C, (c ? v.y : v.x) + hx[m]);
                        pcm_put1(dst + (size_t)(n + 1) * C, (c ? v.w : v.z) + hy[m]);
                    }
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) {
                    const int n = 2 * lane + 128 * t4;
                    if (n < 448) {
                        const dpf4 v = incoming<FROM_LDS, SC1>(p0, p1, n_ch, n);
                        pcm_put1(dst + (size_t)n * C, c ? v.y : v.x);
                        pcm_put1(dst + (size_t)(n + 1) * C, c ? v.w : v.z);
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* run-to-run (and launch-to-launch) rendezvous of the _rv builds (aacg_rv_args, aacg_device.h)  */
/* ------------------------------------------------------------------------------------ */
/* a wave's windowed first half (hx / hy in the IMDCT lane map) as planar [channel][1024] floats in a rendezvous payload:
 * -0.0 where the sequence takes the sample from the overlap alone (EIGHT_SHORT: 0..447), so that tail + head = tail there */
DP_DEVICE void rv_publish_head(float* d0, float* d1, int n_ch, int cls0, int cls1, int lcol,
                               const float (&hx0)[8], const float (&hy0)[8], const float (&hx1)[8], const float (&hy1)[8])
{
    const int lane = dp_lane(), w = lane >> 3, g = lane & 7;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c < n_ch) {
            float* d = c ? d1 : d0;
            const int cls = c ? cls1 : cls0;
            const float (&hx)[8] = c ? hx1 : hx0;
            const float (&hy)[8] = c ? hy1 : hy0;
            if (!cls) {
#pragma unroll
                for (int m = 0; m < 8; m++) dp_g_store_f2(d + 2 * lcol + 128 * m, hx[m], hy[m]);
            } else {
#pragma unroll
                for (int m = 0; m < 8; m++)
                    if (w < 4 || (w == 4 && m < 4)) dp_g_store_f2(d + 448 + 128 * w + 2 * g + 16 * m, hx[m], hy[m]);
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) {
                    const int n = 2 * lane + 128 * t4;
                    if (n < 448) dp_g_store_f2(d + n, -0.0f, -0.0f);
                }
            }
        }
    }
}
/* the tails in a wave's slot (interleaved for two channels) as planar arrays in a rendezvous payload */
DP_DEVICE void rv_publish_tails(const float* slot, int n_ch, float* d0, float* d1)
{
    const int lane = dp_lane();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int n = 2 * lane + 128 * i;
        if (n_ch == 2) {
            const dpf4 a = *(const dpf4*)(slot + 2 * n);           /* (L[n], R[n], L[n+1], R[n+1]) */
            dp_g_store_f2(d0 + n, a.x, a.z);
            dp_g_store_f2(d1 + n, a.y, a.w);
        } else {
            const dpf2 a = *(const dpf2*)(slot + n);
            dp_g_store_f2(d0 + n, a.x, a.y);
        }
    }
}
/* the run that arrives second at a rendezvous whose other side left its windowed first half: the next run's first frame =
 * this wave's tails (its slot) + that first half, to that frame's place in the PCM: `pcm` = sample 0 of the element's first
 * channel there, C = that frame's interleave stride (a frame of this launch, or — pipelined launches — of the next one) */
template <int OUT>
DP_DEVICE void rv_finish_successor(typename pcm_elem<OUT>::type* pcm, int C, int n_ch, const float* slot, const float* h0, const float* h1)
{
#pragma clang fp contract(off)
    const int lane = dp_lane();
    if (n_ch == 2 && C == 2 && ((uintptr_t)pcm & (4u * sizeof(*pcm) - 1u)) == 0) {
        dpf2 hl[8], hr[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const int n = 2 * lane + 128 * i; hl[i] = dp_g_load_f2(h0 + n); hr[i] = dp_g_load_f2(h1 + n); }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int n = 2 * lane + 128 * i;
            const dpf4 t = *(const dpf4*)(slot + 2 * n);
            pcm_put4(pcm + 2 * n, t.x + hl[i].x, t.y + hr[i].x, t.z + hl[i].y, t.w + hr[i].y);
        }
        return;
    }
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const int n = lane + 64 * j;
        if (n_ch == 2) {
            const dpf2 t = *(const dpf2*)(slot + 2 * n);
            pcm_put2(pcm + (size_t)n * C, t.x + dp_g_load_f1(h0 + n), t.y + dp_g_load_f1(h1 + n));
        } else {
            pcm_put1(pcm + (size_t)n * C, slot[n] + dp_g_load_f1(h0 + n));
        }
    }
}

/* overlap buffer r of a channel is 1024 r floats behind its buffer 0; this launch reads buffer (rot + flip) mod AACG_OV_BUFFERS of a
 * run's channel and leaves the new state in the next one (aacg_run.rot, aacg_kparams.flip: both below AACG_OV_BUFFERS, add at most flip + 1) */
DP_DEVICE int ov_buffer(int rot, int add) { int r = rot + add; r = r >= AACG_OV_BUFFERS ? r - AACG_OV_BUFFERS : r; return r >= AACG_OV_BUFFERS ? r - AACG_OV_BUFFERS : r; }

/* ------------------------------------------------------------------------------------ */
/* one run per workgroup                                                                   */
/* ------------------------------------------------------------------------------------ */
/* The run body.  A later run of a chain starts from the tail of the frame before it, which another workgroup owns, so it
 * recomputes that frame's IMDCT.  With up to 15 frames a wave of its own does that; a full run of 16 gives its first
 * wave double duty (DD = true builds only): first the predecessor, whose tails it parks in a scratch area in global
 * memory, then its own frame, which takes its overlap from there.  DD is a template constant, not a run-time flag: the
 * kernels for plans without full later runs (DD = false, aacg_engine.hip) contain no second pass at all — with the pass
 * behind a run-time condition the same source compiled to a run body 0.8 us slower on config 2 (interleaved A/B, 16.8 vs
 * 16.0 us), and a non-inlined predecessor pass cost config 4 more than it saved (29 vs 23 us).  ONE source for both. */
/* EX = true builds (aacg_engine_exrun.hip) carry the optional stages inside the run: noise bands (AACG_PNS_SPEC) in the
 * dequantisation, then the TNS filters (AACG_TNS_SPEC) on the wave's spectra in its own slot before they are staged for
 * the IMDCT — no second kernel and no f32 spectra through HBM (a two-kernel batch costs 26 us before any filter has
 * run, the plain run kernel 14).  They need AACG_RUN_XCH_FLOATS of LDS per wave behind the flags for the TNS carry. */
#define AACG_RUN_XCH_FLOATS AACG_TNS_XCH_FLOATS(AACG_RUN_TNS_ROUND)
#define AACG_LDS_BYTES_F32_EX   (AACG_LDS_BYTES_F32 + 4 * AACG_WG_WAVES * AACG_RUN_XCH_FLOATS)
#define AACG_LDS_BYTES_QUANT_EX (AACG_LDS_BYTES_QUANT + 4 * AACG_WG_WAVES * AACG_RUN_XCH_FLOATS)
/* RV = true builds (aacg_engine_rv.hip): chains longer than a run without a recomputed frame — the plan's runs all start from
 * what the run before them hands over through a rendezvous cell (aacg_rv_args), never from a recomputed predecessor. */
template <int KIND, int OUT = AACG_OUTPUT_F32, bool DD = false, bool EX = false, bool CPL = false, bool RV = false, bool NTL = false, bool PRE = false>
DP_DEVICE void imdct_run_body(const aacg_kparams& P, const aacg_rv_args* V = nullptr, const aacg_run* runs_pre = nullptr, const aacg_tables* tab_pre = nullptr, const aacg_rv_link* links_pre = nullptr,
                              const aacg_dev_unit* units_pre = nullptr, const void* coeffs_pre = nullptr, const aacg_band_meta* meta_pre = nullptr)
{
    const aacg_run* const k_runs = PRE ? runs_pre : P.runs;
    const aacg_tables* const k_tab = PRE ? tab_pre : P.tab;
    const aacg_dev_unit* const k_units = PRE ? units_pre : P.units;
    const void* const k_coeffs = PRE ? coeffs_pre : P.coeffs;
    const aacg_band_meta* const k_meta = PRE ? meta_pre : P.meta;
    const int TAB_FLOATS = (KIND == AACG_INPUT_QUANT_I16) ? AACG_TAB_QUANT_FLOATS : AACG_TAB_F32_FLOATS;
    const int lane = dp_lane(), wave = dp_wave();
    const aacg_run* run = k_runs + dp_block();
    float* lds = (float*)dp_lds_fixed<4 * AACG_LDS_FLOATS(TAB_FLOATS) + (EX ? 4 * AACG_WG_WAVES * AACG_RUN_XCH_FLOATS : 0)>();
    const float* tab = lds;
    float* slots = lds + AACG_TAB_SLOT_BASE(TAB_FLOATS);
    float* slot = slots + wave * AACG_SLOT_FLOATS;
    int* flags = (int*)(slots + AACG_WG_WAVES * AACG_SLOT_FLOATS);
    float* xch = EX ? (float*)(flags + AACG_WG_WAVES) + wave * AACG_RUN_XCH_FLOATS : nullptr;

    /* The start of a run is dependent memory round trips and nothing else (aacg_run, aacg_device.h): behind the kernel
     * arguments ONE batch — the run record's header, the link record, the three words of the record that belong to this wave
     * (its unit, where that unit's spectra and band words lie), and the table loads, which stay in flight across the wait —
     * then the spectra's requests and, beside them, the unit record.  Spelled out (dp_sload / dp_swait): left to itself hipcc
     * makes a chain of five round trips out of the same reads. */
    dp_su8 rh = dp_sload8(run, k_tab);                 /* (the table pointer with the first batch of kernel arguments) */
    dp_su4 lkw = dp_sload4(RV ? (const void*)((PRE ? links_pre : V->links) + dp_block()) : (const void*)run);
    unsigned w_unit = dp_sload1(run, (int)__builtin_offsetof(aacg_run, wave_unit) + 4 * wave);
    unsigned w_coef = dp_sload1(run, (int)__builtin_offsetof(aacg_run, wave_coef) + 4 * wave);
    unsigned w_meta = dp_sload1(run, (int)__builtin_offsetof(aacg_run, wave_meta) + 4 * wave);
    dpf4 tr0, tr1;
    stage_tables_load(k_tab, TAB_FLOATS, tr0, tr1);
    dp_swait(rh, lkw, w_unit, w_coef, w_meta);

    const int pred_unit = (int)rh[0], n_units = (int)rh[1];
    const bool run_is_last = rh[2] != 0;
    const int w_nch = (int)((rh[3] >> (2 * wave)) & 3u);
    const int run_ov0[2] = {(int)rh[4], (int)rh[5]}, run_rot[2] = {(int)rh[6], (int)rh[7]};
    const bool has_pred = !RV && pred_unit >= 0;
    aacg_rv_link lk; lk.link_in = lk.link_out = lk.succ_unit = -1; lk.reserved = 0;
    if (RV) { lk.link_in = (int)lkw[0]; lk.link_out = (int)lkw[1]; lk.succ_unit = (int)lkw[2]; }
    /* pipelined launches (aacg_decode_pipelined): the two ends of a chain meet the plan's neighbouring LAUNCHES in cross-launch
     * cells (aacg_xl_cell) exactly as its runs meet each other in the in-launch ones */
    const bool xl = RV && V->xl_cells != nullptr;
    /* the frame another workgroup may be waiting for: a run's last, when its chain goes on — in this launch or in the next */
    const bool hands_over = RV && wave == n_units - 1 && (lk.link_out >= 0 || xl);
    const bool hurry = hands_over;
    /* this launch's overlap buffers of the run's channels (float offsets in the pool), evaluated where a chain's first or last wave needs them */
#define AACG_OV_IN(c)  (run_ov0[c] + 1024 * ov_buffer(run_rot[c], P.flip))
#define AACG_OV_OUT(c) (run_ov0[c] + 1024 * ov_buffer(run_rot[c], P.flip + 1))
    /* A later run of a chain starts from the tail of the frame before it, which another workgroup owns, so it
     * recomputes that frame's IMDCT.  With up to 15 frames wave 0 does only that (waves 1.. own the frames);
     * a full run of 16 frames gives wave 0 double duty: first the predecessor (its tail goes to a scratch
     * area in global memory), then its own frame, which takes its overlap from that scratch area.
     * (aacg_run_wave_unit is the same rule on the planner's side: wave_unit / wave_coef / wave_meta are per WAVE.) */
    const bool dd = DD && has_pred && n_units == AACG_WG_WAVES;
    const bool is_pred_wave = has_pred && !dd && wave == 0;
    const bool active = (has_pred && !dd) ? (wave == 0 || wave - 1 < n_units) : wave < n_units;     /* this wave has a frame */
    const int n_pass = (DD && dd && wave == 0) ? 2 : 1;
    float* scratch = DD ? P.scratch + (size_t)dp_block() * AACG_SLOT_FLOATS : nullptr;

    float hx0[8], hy0[8], hx1[8], hy1[8];
    /* the unit record (a wave without a frame reads unit 0's and ignores it): scalar loads — nothing that may clobber memory
     * (stores, clock reads) precedes them — requested here, first used behind the table barrier */
    unit_view u = load_unit(k_units + dp_uniform((int)w_unit));
    /* Earlier frames get the higher issue priority: they finish first and their PCM stores overlap
     * the later waves' arithmetic.  A wave only ever waits for the wave before it, whose priority is
     * never lower, so a spinning consumer cannot starve its producer. */
    /* (the multichannel variants leave every wave at the default priority: 62.0 -> 61.9 us on config 5, nothing anywhere else) */
    if (NTL && !RV) {} else
    if (AACG_ABL(P, 64)) dp_setprio(0); else if (AACG_ABL(P, 32)) dp_setprio(1 - (wave >> 3)); else dp_setprio(hurry ? 3 : 3 - (wave >> 2));
    const unsigned long long t_start = AACG_ABL(P, 16) ? dp_clock() : 0;
    /* (the coupling builds carry their side buffer in spec_out, aacg_set_cpl: never a trace there) */
    unsigned long long* trace = (!CPL && AACG_ABL(P, 16)) ? (unsigned long long*)P.spec_out + ((size_t)dp_block() * AACG_WG_WAVES + wave) * 8 : nullptr;
    if (trace && lane == 0) trace[0] = t_start;

    /* This wave's spectrum: only the tables are waited for before the barrier.  All loads are
     * unconditional (an idle wave of a short run re-reads unit 0 and ignores it; a single channel reads
     * its block twice): a load under a condition makes hipcc wait vmcnt(0) at the join, which would
     * serialise the HBM round trips. */
    quant_regs qreg;
    dpf4 xa[4], xb[4];
    int n_ch = 0, cls0 = 0, cls1 = 0;                  /* from the unit record, whose first use is behind the table barrier */
    bool pair_path = false;
    auto issue_loads = [&](uint32_t coef_block, uint32_t meta_block, int nch) {
        if (KIND == AACG_INPUT_QUANT_I16) quant_load<NTL>(k_coeffs, k_meta, coef_block, meta_block, nch, qreg);
        else {
            const float* xsrc = (const float*)k_coeffs + (size_t)coef_block * 1024u;
            const float* xsrc1 = xsrc + (nch == 2 ? 1024 : 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (NTL) { xa[i] = dp_load_nt((const dpf4*)(xsrc + 4 * lane + 256 * i)); xb[i] = dp_load_nt((const dpf4*)(xsrc1 + 4 * lane + 256 * i)); }
                else     { xa[i] = *(const dpf4*)(xsrc + 4 * lane + 256 * i); xb[i] = *(const dpf4*)(xsrc1 + 4 * lane + 256 * i); }
            }
        }
    };
    /* the tails of a unit from its slot (interleaved for two channels) to planar arrays in global memory */
    auto save_tails = [&](float* d0, float* d1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = 4 * lane + 256 * i;
            if (n_ch == 2) {
                const dpf4 a = *(const dpf4*)(slot + 2 * n), b = *(const dpf4*)(slot + 2 * n + 4);
                dpf4 l4, r4;
                l4.x = a.x; l4.y = a.z; l4.z = b.x; l4.w = b.z;
                r4.x = a.y; r4.y = a.w; r4.z = b.y; r4.w = b.w;
                *(dpf4*)(d0 + n) = l4;
                *(dpf4*)(d1 + n) = r4;
            } else {
                *(dpf4*)(d0 + n) = *(const dpf4*)(slot + n);
            }
        }
    };
    /* Load staggering (measured: -1 us on the f32 path).  The first four frames of the run (the highest-
     * priority waves, one per SIMD; two on the int16 path, where a sweep of 0..8 early waves at steady clocks
     * gave 14.5 / 14.2 / 13.5 / 13.9 / 14.1 / - / - / - / 14.8 us) request their spectra first and alone: the table barrier below is only
     * released once their data has landed (the loads sit under a condition, so hipcc waits for them at the
     * join), and only then do the other twelve waves issue their requests.  The first group therefore sees
     * its data after ~1.7 us instead of queueing behind the whole chip's 33 MB, and the later groups' data
     * arrives while the SIMD is still busy with the earlier ones. */
    const bool early = wave < (KIND == AACG_INPUT_QUANT_I16 ? 2 : 4) || hurry || AACG_ABL(P, 128);
    if (early) issue_loads(w_coef, w_meta, w_nch);
    stage_tables_store(lds, TAB_FLOATS, tr0, tr1);
    if (lane == 0) flags[wave] = 0;
    dp_block_sync_lds();                               /* tables and flags are in LDS */
    if (!early) issue_loads(w_coef, w_meta, w_nch);
    if (trace && lane == 0) trace[1] = dp_clock();
    n_ch = active ? u.n_ch : 0;
    cls0 = u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE;
    cls1 = u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE;
    pair_path = n_ch == 2 && u.seq[0] == u.seq[1] && u.shape[0] == u.shape[1] && u.shape_prev[0] == u.shape_prev[1];

    /* dequantise / stage this wave's loaded spectrum and run the filterbank on it: tails into the slot, the
     * windowed first half into hx / hy */
    auto front = [&](bool want_head) {
        if (trace) { dp_vm_drain(); if (lane == 0) trace[6] = dp_clock(); }     /* profiling: this wave's loads have landed */
        if (AACG_ABL(P, 1) && KIND != AACG_INPUT_QUANT_I16) {
#pragma unroll
            for (int m = 0; m < 8; m++) { hx0[m] = xa[m & 3].x; hy0[m] = xa[m & 3].y; hx1[m] = xb[m & 3].z; hy1[m] = xb[m & 3].w; }
            return;
        }
        if (KIND == AACG_INPUT_QUANT_I16) {
            float xl[16], xr[16];
            if (AACG_ABL(P, 8)) {                        /* profiling: no dequant / MS / IS arithmetic */
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    xl[8 * i] = (float)qreg.ql[i].x; xl[8 * i + 1] = (float)qreg.ql[i].y; xl[8 * i + 2] = (float)qreg.ql[i].z; xl[8 * i + 3] = (float)qreg.ql[i].w;
                    xl[8 * i + 4] = xl[8 * i]; xl[8 * i + 5] = xl[8 * i + 1]; xl[8 * i + 6] = xl[8 * i + 2]; xl[8 * i + 7] = xl[8 * i + 3];
                    xr[8 * i] = (float)qreg.qr[i].x; xr[8 * i + 1] = (float)qreg.qr[i].y; xr[8 * i + 2] = (float)qreg.qr[i].z; xr[8 * i + 3] = (float)qreg.qr[i].w;
                    xr[8 * i + 4] = xr[8 * i]; xr[8 * i + 5] = xr[8 * i + 1]; xr[8 * i + 6] = xr[8 * i + 2]; xr[8 * i + 7] = xr[8 * i + 3];
                }
            } else
            spectral_quant<EX>(P, tab, u, n_ch, qreg, slot + 1024, xl, xr);
            /* TNS runs here (decoder.js:309-313): identity as the reference executes it (tns.js:106,122), so the plain
             * kernels have nothing to do; the EX builds apply the filters AACG_TNS_SPEC asks for */
            if (EX && P.tns && (u.tns[0] || (n_ch == 2 && u.tns[1]))) {
                stage_nat8(xl, slot);
                if (n_ch == 2) stage_nat8(xr, slot + 1024);
                dp_wave_sync();
                tns_unit<AACG_RUN_TNS_ROUND>(slot, xch, u.tns[0] ? P.tns + u.tns_offset : nullptr, (n_ch == 2 && u.tns[1]) ? P.tns + u.tns_offset + 1 : nullptr,
                                             u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE, u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE, P.tns, aacg_tns_m(P));
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const dpf4 a = *(const dpf4*)(slot + 8 * lane + 512 * i), b = *(const dpf4*)(slot + 8 * lane + 512 * i + 4);
                    xl[8 * i] = a.x; xl[8 * i + 1] = a.y; xl[8 * i + 2] = a.z; xl[8 * i + 3] = a.w;
                    xl[8 * i + 4] = b.x; xl[8 * i + 5] = b.y; xl[8 * i + 6] = b.z; xl[8 * i + 7] = b.w;
                    /* unconditionally (a single channel reads what nobody uses): with the read under n_ch == 2 the old xr
                     * stays live across the filters and goes to scratch (18 MB per 4096-frame batch, written and read) */
                    const dpf4 c = *(const dpf4*)(slot + 1024 + 8 * lane + 512 * i), d = *(const dpf4*)(slot + 1024 + 8 * lane + 512 * i + 4);
                    xr[8 * i] = c.x; xr[8 * i + 1] = c.y; xr[8 * i + 2] = c.z; xr[8 * i + 3] = c.w;
                    xr[8 * i + 4] = d.x; xr[8 * i + 5] = d.y; xr[8 * i + 6] = d.z; xr[8 * i + 7] = d.w;
                }
                dp_wave_sync();
            }
            if (pair_path) stage_pair_nat8(xl, xr, slot);
            else { stage_nat8(xl, slot); if (n_ch == 2) stage_nat8(xr, slot + 1024); }
        } else {
            if (EX && P.tns && (u.tns[0] || (n_ch == 2 && u.tns[1]))) {
                /* the spectra as loaded are in ICStream.data order already: through the slot and back */
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    *(dpf4*)(slot + 4 * lane + 256 * i) = xa[i];
                    if (n_ch == 2) *(dpf4*)(slot + 1024 + 4 * lane + 256 * i) = xb[i];
                }
                dp_wave_sync();
                tns_unit<AACG_RUN_TNS_ROUND>(slot, xch, u.tns[0] ? P.tns + u.tns_offset : nullptr, (n_ch == 2 && u.tns[1]) ? P.tns + u.tns_offset + 1 : nullptr,
                                             u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE, u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE, P.tns, aacg_tns_m(P));
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    xa[i] = *(const dpf4*)(slot + 4 * lane + 256 * i);
                    xb[i] = *(const dpf4*)(slot + (n_ch == 2 ? 1024 : 0) + 4 * lane + 256 * i);
                }
                dp_wave_sync();
            }
            if (pair_path) stage_pair_f32(xa, xb, slot);
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    *(dpf4*)(slot + 4 * lane + 256 * i) = xa[i];
                    if (n_ch == 2) *(dpf4*)(slot + 1024 + 4 * lane + 256 * i) = xb[i];
                }
            }
        }
        dp_wave_sync();
        if (trace && lane == 0) trace[2] = dp_clock();     /* spectrum arrived and staged */
        filter_unit<AACG_VM_KIND(KIND)>(tab, u, n_ch, pair_path, want_head, slot, hx0, hy0, hx1, hy1);
    };

    if (active) front(!is_pred_wave && n_pass == 1);
    if (DD && n_pass == 2) {
        /* double duty (cold: only the first wave of a full later run): park the predecessor's tails, then fetch
         * and process this wave's own frame with a second copy of the code above */
        dp_keep_branch();
        dp_wave_sync();
        save_tails(scratch, scratch + 1024);
        u = load_unit(P.units + dp_uniform(run->unit[0]));
        n_ch = u.n_ch;
        cls0 = u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE;
        cls1 = u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE;
        pair_path = n_ch == 2 && u.seq[0] == u.seq[1] && u.shape[0] == u.shape[1] && u.shape_prev[0] == u.shape_prev[1];
        issue_loads(u.coef_offset, u.meta_offset, u.n_ch);
        dp_wave_sync();                                /* the slot is free again */
        front(true);
    }

    /* this wave's tails are complete in its slot: release them to the next wave */
    dp_wave_sync();
    if (lane == 0) dp_flag_set(&flags[wave], 1);
    if (trace && lane == 0) trace[3] = dp_clock();         /* IMDCT done, tail released */

    typedef typename pcm_elem<OUT>::type pcm_t;
    const unsigned long long rv_tag = RV ? V->epoch << 2 : 0ull;
    if (RV && hands_over && active) {
        /* the chain goes on in another workgroup: publish this frame's tails — or, if that workgroup was here first and left
         * its windowed first half, finish its frame (nobody waits for anybody: no dispatch order is assumed).  `cross`: the
         * other workgroup belongs to the plan's NEXT launch; the tails then land in the out overlap buffer itself */
        dp_keep_branch();
        const bool cross = lk.link_out < 0;
        aacg_xl_cell* cell = cross ? V->xl_cells + ((run_ov0[0] >> 10) + ov_buffer(run_rot[0], P.flip + 1)) : nullptr;
        float* data = cross ? nullptr : V->data + (size_t)lk.link_out * AACG_RV_DATA_FLOATS;
        unsigned long long* st = cross ? &cell->state : V->state + (size_t)lk.link_out * AACG_RV_STATE_WORDS;
        float* t0 = cross ? P.overlap + AACG_OV_OUT(0) : data;
        float* t1 = cross ? P.overlap + AACG_OV_OUT(1) : data + 1024;
        const float* h0 = cross ? V->xl_head + AACG_OV_OUT(0) : data + 2048;
        const float* h1 = cross ? V->xl_head + AACG_OV_OUT(1) : data + 2048 + 1024;
        const unsigned long long seen = dp_first_u64(dp_g_load_u64(st));
        bool theirs = seen == (rv_tag | AACG_RV_HEAD);
        if (!theirs) {
            rv_publish_tails(slot, n_ch, t0, t1);
            dp_vm_drain();
            dp_wave_sync();
            bool won = true;
            if (lane == 0) won = dp_g_cas_u64(st, seen, rv_tag | AACG_RV_TAIL);
            theirs = dp_first_u64(won ? 0ull : 1ull) != 0ull;
        }
        if (theirs) {
            pcm_t* dst; int C;
            if (cross) {
                dst = (pcm_t*)(uintptr_t)dp_first_u64(dp_g_load_u64(&cell->pcm));
                C = dp_uniform((int)dp_g_load_u32(&cell->n_out_ch));
            } else {
                const unit_view su = load_unit(P.units + dp_uniform(lk.succ_unit));
                dst = (pcm_t*)P.pcm + su.pcm_offset + su.channel;
                C = su.n_out_ch;
            }
            rv_finish_successor<OUT>(dst, C, n_ch, slot, h0, h1);
        }
    }

    if (active && !is_pred_wave && AACG_ABL(P, 2)) {
        /* profiling: keep the values live without storing 8 KiB of PCM */
        float acc = 0.0f;
#pragma unroll
        for (int m = 0; m < 8; m++) acc += hx0[m] + hy0[m] + hx1[m] + hy1[m];
        if (acc == 123456.789f) P.pcm[0] = acc;
    } else if (active && !is_pred_wave) {
        /* the long paths of the int16 seam deal their columns out by long_col (filter_unit<VM>) */
        const int lcol = AACG_VM_KIND(KIND) ? long_col(lane) : lane;
        if (wave == 0) {
            /* first frame of its chain in this launch: overlap state from HBM (filter_bank.js:38-41,
             * `overlap = this.overlaps[channel]`); a double-duty wave: the tails it parked itself */
            const float* ov0 = n_pass == 2 ? scratch : P.overlap + AACG_OV_IN(0);
            const float* ov1 = n_pass == 2 ? scratch + 1024 : P.overlap + AACG_OV_IN(1);
            cpl_prefetch none; none.n = 0;
            /* first frame of its chain in a pipelined launch whose predecessor may still be running: its state arrives through
             * the cross-launch cell of the in buffer (epoch_in = 0: the predecessor is known to be complete, a plain read) */
            const bool cross = xl && lk.link_in < 0 && V->epoch_in != 0ull;
            if (RV && (lk.link_in >= 0 || cross)) {
                /* the tails the run before it published — or, if they are not there yet, leave the windowed first half for that
                 * run (and, across launches, where the finished samples go) and go */
                dp_keep_branch();
                aacg_xl_cell* cell = cross ? V->xl_cells + ((run_ov0[0] >> 10) + ov_buffer(run_rot[0], P.flip)) : nullptr;
                float* data = cross ? nullptr : V->data + (size_t)lk.link_in * AACG_RV_DATA_FLOATS;
                unsigned long long* st = cross ? &cell->state : V->state + (size_t)lk.link_in * AACG_RV_STATE_WORDS;
                const float* t0 = cross ? P.overlap + AACG_OV_IN(0) : data;
                const float* t1 = cross ? P.overlap + AACG_OV_IN(1) : data + 1024;
                float* h0 = cross ? V->xl_head + AACG_OV_IN(0) : data + 2048;
                float* h1 = cross ? V->xl_head + AACG_OV_IN(1) : data + 2048 + 1024;
                const unsigned long long tag = cross ? V->epoch_in << 2 : rv_tag;
                const unsigned long long seen = dp_first_u64(dp_g_load_u64(st));
                bool theirs = seen == (tag | AACG_RV_TAIL);
                if (!theirs) {
                    rv_publish_head(h0, h1, n_ch, cls0, cls1, lcol, hx0, hy0, hx1, hy1);
                    if (cross && lane == 0) {
                        dp_g_store_u64(&cell->pcm, (unsigned long long)(uintptr_t)((pcm_t*)P.pcm + u.pcm_offset + u.channel));
                        dp_g_store_u32(&cell->n_out_ch, (unsigned)u.n_out_ch);
                    }
                    dp_vm_drain();
                    dp_wave_sync();
                    bool won = true;
                    if (lane == 0) won = dp_g_cas_u64(st, seen, tag | AACG_RV_HEAD);
                    theirs = dp_first_u64(won ? 0ull : 1ull) != 0ull;
                }
                if (theirs) epilogue<false, OUT, CPL, true>(P, none, t0, t1, u, n_ch, cls0, cls1, P.pcm, hx0, hy0, hx1, hy1, lcol);
            } else
            epilogue<false, OUT, CPL>(P, none, ov0, ov1, u, n_ch, cls0, cls1, P.pcm, hx0, hy0, hx1, hy1, lcol);
        } else {
            cpl_prefetch pre;
            couple_prefetch<CPL>(P, u, pre);
            dp_flag_wait(&flags[wave - 1], 1);         /* the previous frame's tails (acquire) */
            if (trace && lane == 0) trace[4] = dp_clock();
            epilogue<true, OUT, CPL>(P, pre, slot - AACG_SLOT_FLOATS, slot - AACG_SLOT_FLOATS, u, n_ch, cls0, cls1, P.pcm, hx0, hy0, hx1, hy1, lcol);
        }
        if (trace && lane == 0) trace[5] = dp_clock();     /* PCM stores issued */
        /* the chain's last frame in this launch: its tail is the new overlap state (planar in HBM) */
        const int last_wave = (has_pred && !dd) ? n_units : n_units - 1;
        if (wave == last_wave && run_is_last && !xl)      /* (pipelined: the hand-over above has put it there, or the next launch has taken it) */
            save_tails(P.overlap + AACG_OV_OUT(0), P.overlap + AACG_OV_OUT(1));
    }
    if (trace) {                                           /* profiling: when this wave's stores were acknowledged (a wave ends no earlier), and on which CU it ran */
        dp_vm_drain();
        if (lane == 0) trace[7] = ((unsigned long long)dp_cu_id() << 52) | (dp_clock() & 0xfffffffffffffull);
    }
}

#undef AACG_OV_IN
#undef AACG_OV_OUT

/* The optional stages as a kernel of their own (16 units per workgroup, one wave each), so that the run kernels
 * never carry them: quantised input -> dequantisation, noise bands (AACG_PNS_SPEC), MS / IS, then the TNS filters
 * (AACG_TNS_SPEC); f32 input -> the TNS filters only.  f32 spectra in ICStream.data order to spec_out, which the
 * f32 run kernel then consumes.  LDS: the dequantisation part of the tables only (scalefactors, IQ, band maps: `tab` is
 * biased so that the usual offsets work), then per wave a 2048-float slot (band records and PNS scratch first; for TNS
 * both channels' spectra) and the TNS exchange area. */
#define AACG_SPX_TAB_FLOATS  (AACG_TAB_QUANT_FLOATS - AACG_TAB_F32_FLOATS)
#define AACG_SPX_TNS_ROUND   4      /* rounds of 8 cost 12 bytes per lane of scratch */
#define AACG_SPX_XCH_FLOATS  AACG_TNS_XCH_FLOATS(AACG_SPX_TNS_ROUND)
#define AACG_SPX_WAVE_FLOATS (AACG_SLOT_FLOATS + AACG_SPX_XCH_FLOATS)
template <int KIND>
DP_DEVICE void spectral_ex_body(const aacg_kparams& P, int n_units)
{
    const int TAB_FLOATS = (KIND == AACG_INPUT_QUANT_I16) ? AACG_SPX_TAB_FLOATS : 0;
    const int lane = dp_lane(), wave = dp_wave();
    float* lds = (float*)dp_lds();
    const float* tab = lds - AACG_TAB_F32_FLOATS;      /* only offsets >= AACG_TAB_OFF_SF are ever read through it */
    float* slot = lds + TAB_FLOATS + wave * AACG_SPX_WAVE_FLOATS;
    float* xch = slot + AACG_SLOT_FLOATS;
    if (KIND == AACG_INPUT_QUANT_I16) {
        stage_tables((const aacg_tables*)((const float*)P.tab + AACG_TAB_F32_FLOATS), lds, AACG_SPX_TAB_FLOATS);
        dp_block_sync();
    }
    const int ui = dp_block() * AACG_WG_WAVES + wave;
    if (ui >= n_units) return;
    const unit_view u = load_unit(P.units + dp_uniform(ui));
    const int n_ch = u.n_ch;
    float xl[16], xr[16];
    if (KIND == AACG_INPUT_QUANT_I16) {
        quant_regs qreg;
        quant_load(P.coeffs, P.meta, u.coef_offset, u.meta_offset, n_ch, qreg);
        spectral_quant<true>(P, tab, u, n_ch, qreg, slot, xl, xr);
    } else {
        const float* x0 = (const float*)P.coeffs + (size_t)u.coef_offset * 1024u;
        const float* x1 = x0 + (n_ch == 2 ? 1024 : 0);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const dpf4 a = *(const dpf4*)(x0 + 8 * lane + 512 * i), b = *(const dpf4*)(x0 + 8 * lane + 512 * i + 4);
            const dpf4 c = *(const dpf4*)(x1 + 8 * lane + 512 * i), d = *(const dpf4*)(x1 + 8 * lane + 512 * i + 4);
            xl[8 * i] = a.x; xl[8 * i + 1] = a.y; xl[8 * i + 2] = a.z; xl[8 * i + 3] = a.w;
            xl[8 * i + 4] = b.x; xl[8 * i + 5] = b.y; xl[8 * i + 6] = b.z; xl[8 * i + 7] = b.w;
            xr[8 * i] = c.x; xr[8 * i + 1] = c.y; xr[8 * i + 2] = c.z; xr[8 * i + 3] = c.w;
            xr[8 * i + 4] = d.x; xr[8 * i + 5] = d.y; xr[8 * i + 6] = d.z; xr[8 * i + 7] = d.w;
        }
    }
    /* tns.process (decoder.js:309-313) as it was meant to run, after MS / IS: both channels side by side */
    const bool tns0 = P.tns && u.tns[0], tns1 = P.tns && n_ch == 2 && u.tns[1];
    if (tns0 || tns1) {
        stage_nat8(xl, slot);
        if (n_ch == 2) stage_nat8(xr, slot + 1024);
        dp_wave_sync();
        tns_unit<AACG_SPX_TNS_ROUND>(slot, xch, tns0 ? P.tns + u.tns_offset : nullptr, tns1 ? P.tns + u.tns_offset + 1 : nullptr,
                 u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE, u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE, P.tns, aacg_tns_m(P));
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const dpf4 a = *(const dpf4*)(slot + 8 * lane + 512 * i), b = *(const dpf4*)(slot + 8 * lane + 512 * i + 4);
            xl[8 * i] = a.x; xl[8 * i + 1] = a.y; xl[8 * i + 2] = a.z; xl[8 * i + 3] = a.w;
            xl[8 * i + 4] = b.x; xl[8 * i + 5] = b.y; xl[8 * i + 6] = b.z; xl[8 * i + 7] = b.w;
            /* unconditionally: see the same read in imdct_run_body */
            const dpf4 c = *(const dpf4*)(slot + 1024 + 8 * lane + 512 * i), d = *(const dpf4*)(slot + 1024 + 8 * lane + 512 * i + 4);
            xr[8 * i] = c.x; xr[8 * i + 1] = c.y; xr[8 * i + 2] = c.z; xr[8 * i + 3] = c.w;
            xr[8 * i + 4] = d.x; xr[8 * i + 5] = d.y; xr[8 * i + 6] = d.z; xr[8 * i + 7] = d.w;
        }
    }
    float* out = P.spec_out + (size_t)u.coef_offset * 1024u;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        dpf4 a, b;
        a.x = xl[8 * i]; a.y = xl[8 * i + 1]; a.z = xl[8 * i + 2]; a.w = xl[8 * i + 3];
        b.x = xl[8 * i + 4]; b.y = xl[8 * i + 5]; b.z = xl[8 * i + 6]; b.w = xl[8 * i + 7];
        *(dpf4*)(out + 8 * lane + 512 * i) = a;
        *(dpf4*)(out + 8 * lane + 512 * i + 4) = b;
        if (n_ch == 2) {
            a.x = xr[8 * i]; a.y = xr[8 * i + 1]; a.z = xr[8 * i + 2]; a.w = xr[8 * i + 3];
            b.x = xr[8 * i + 4]; b.y = xr[8 * i + 5]; b.z = xr[8 * i + 6]; b.w = xr[8 * i + 7];
            *(dpf4*)(out + 1024 + 8 * lane + 512 * i) = a;
            *(dpf4*)(out + 1024 + 8 * lane + 512 * i + 4) = b;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* AACG_CCE_SPEC: coupling channel elements (cce.js:121-158, decoder.js:406-433 as meant)    */
/* ------------------------------------------------------------------------------------ */
/* Dependent coupling, one wave per (coupling element, target channel) job: target[k] += gain[band of k] * cce[k] over
 * the coupling element's coded bands (cce.js:130-158; the band of a coefficient comes from the same maps the
 * dequantisation uses, the group of a short window from the element's group map).  One rounding per sample, like the
 * reference's `data[i] += gain * iqData[i]` on a Float32Array: a fused multiply-add. */
DP_DEVICE void couple_spec_body(const aacg_couple_params& Q, int waves_per_block)
{
    const int lane = dp_lane(), j = dp_block() * waves_per_block + dp_wave();
    if (j >= Q.n_jobs) return;
    const aacg_couple_job& job = Q.jobs[j];
    const unit_view u = load_unit(Q.units + dp_uniform((int)job.cce_unit));
    const int is_short = u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE, max_sfb = u.max_sfb[0];
    const float* src = Q.spec + (size_t)job.src * 1024u;
    float* dst = Q.spec + (size_t)job.dst * 1024u;
    const float* gains = Q.gains + job.gain_off;
    const aacg_band_meta* m = Q.meta ? Q.meta + u.meta_offset : nullptr;
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int k = lane + 64 * i;
        const int sfb = is_short ? Q.tab->band_of_short[k & 127] : Q.tab->band_of_long[k];
        const int g = is_short ? (int)((u.gmap[0] >> (4 * (k >> 7))) & 15u) : 0;
        if (sfb < max_sfb) {
            const int idx = g * max_sfb + sfb;
            if (!m || (m->band[idx] >> AACG_META_BT_SHIFT) != AACG_ZERO_BT) dst[k] = dp_fma(gains[idx], src[k], dst[k]);
        }
    }
}

/* Independent coupling: the coupling element's own filterbank output (side buffer, PCM-scaled like everything the
 * windows touched) times the list's gain, added to the target channel's samples in the interleaved PCM. */
DP_DEVICE void couple_pcm_body(const aacg_couple_params& Q, int waves_per_block)
{
    const int lane = dp_lane(), j = dp_block() * waves_per_block + dp_wave();
    if (j >= Q.n_jobs) return;
    const aacg_couple_job& job = Q.jobs[j];
    const float gain = Q.gains[job.gain_off];
    const float* src = Q.side + (size_t)job.src * 1024u;
    float* dst = Q.pcm + job.dst;
    const size_t stride = job.stride;
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int n = lane + 64 * i;
        dst[n * stride] = dp_fma(gain, src[n], dst[n * stride]);
    }
}

/* Spectral stage alone (16 units per workgroup, one wave each): spec_out in ICStream.data order. */
DP_DEVICE void spectral_body(const aacg_kparams& P, int n_units)
{
    const int lane = dp_lane(), wave = dp_wave();
    float* lds = (float*)dp_lds();
    const float* tab = lds;
    float* bt = lds + AACG_TAB_QUANT_FLOATS + wave * 512;
    stage_tables(P.tab, lds, AACG_TAB_QUANT_FLOATS);
    dp_block_sync();
    const int ui = dp_block() * AACG_WG_WAVES + wave;
    if (ui >= n_units) return;
    const unit_view u = load_unit(P.units + dp_uniform(ui));
    const int n_ch = u.n_ch;
    float xl[16], xr[16];
    quant_regs qreg;
    quant_load(P.coeffs, P.meta, u.coef_offset, u.meta_offset, n_ch, qreg);
    spectral_quant(P, tab, u, n_ch, qreg, bt, xl, xr);
    float* out = P.spec_out + (size_t)u.coef_offset * 1024u;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        dpf4 a, b;
        a.x = xl[8 * i]; a.y = xl[8 * i + 1]; a.z = xl[8 * i + 2]; a.w = xl[8 * i + 3];
        b.x = xl[8 * i + 4]; b.y = xl[8 * i + 5]; b.z = xl[8 * i + 6]; b.w = xl[8 * i + 7];
        *(dpf4*)(out + 8 * lane + 512 * i) = a;
        *(dpf4*)(out + 8 * lane + 512 * i + 4) = b;
        if (n_ch == 2) {
            a.x = xr[8 * i]; a.y = xr[8 * i + 1]; a.z = xr[8 * i + 2]; a.w = xr[8 * i + 3];
            b.x = xr[8 * i + 4]; b.y = xr[8 * i + 5]; b.z = xr[8 * i + 6]; b.w = xr[8 * i + 7];
            *(dpf4*)(out + 1024 + 8 * lane + 512 * i) = a;
            *(dpf4*)(out + 1024 + 8 * lane + 512 * i + 4) = b;
        }
    }
}

#endif /* AACG_KERNELS_H */
