/*
 * aacg_kernels.h — the hot path as wavefront code for CDNA4 (gfx950).
 *
 * Work decomposition:
 *   wave       = one unit (SCE/LFE/CPE) of one frame: both channels of a CPE live in one
 *                wave, so MS/IS are register-local and stereo PCM leaves as 16-byte
 *                (L,R,L,R) stores, 1 KiB contiguous per wave instruction.
 *   workgroup  = one run: up to AACG_RUN_W consecutive frames of that element + wave 0,
 *                which supplies the tail feeding the first frame (recomputed from the
 *                previous frame's spectrum, or copied from the overlap state in HBM).
 *                Tails travel wave -> wave through LDS; one workgroup barrier per run.
 *   IMDCT      = N/4-point complex inverse FFT between two twiddle passes (the algorithm
 *                class of mdct.js:62-115), here as radix-8 register butterflies with two
 *                LDS transposes (512 = 8x8x8) or one (64 = 8x8 per short window), natural
 *                order in and out, so no bit-reversal pass (fft.js:113-137) exists at all.
 *
 * Lane maps.  Long: lane l, element j holds index l + 64 j.  Short: lane (w = l>>3, g = l&7),
 * element j holds index g + 8 j of window w.  Both FFTs return to the map they start in,
 * so pre- and post-twiddle use the same sincos registers.
 */
#ifndef AACG_KERNELS_H
#define AACG_KERNELS_H

#include "devport.h"
#include "aacg_device.h"

struct cpx { float re, im; };

DP_DEVICE cpx c_add(cpx a, cpx b) { cpx r; r.re = a.re + b.re; r.im = a.im + b.im; return r; }
DP_DEVICE cpx c_sub(cpx a, cpx b) { cpx r; r.re = a.re - b.re; r.im = a.im - b.im; return r; }
DP_DEVICE cpx c_muli(cpx a)       { cpx r; r.re = -a.im; r.im = a.re; return r; }            /* i * a */
DP_DEVICE cpx c_mul(cpx a, aacg_c2 w)
{
    cpx r;
    r.re = dp_fma(a.re, w.re, -(a.im * w.im));
    r.im = dp_fma(a.re, w.im, a.im * w.re);
    return r;
}

/* 8-point inverse DFT  y[q] = sum_j x[j] e^{+2 pi i j q / 8}, in place. */
DP_DEVICE void radix8_inv(cpx (&x)[8])
{
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

/* array position of the (even, odd) coefficient pair held by (lane, j) */
DP_DEVICE int pair_pos(int lane, int j, int cls)
{
    return cls ? (((lane >> 3) << 7) + ((lane & 7) << 1) + (j << 4))      /* short: 128 w + 2 g + 16 j */
               : ((lane << 1) + (j << 7));                                /* long : 2 l + 128 j        */
}

/* ------------------------------------------------------------------------------------ */
/* long windows: IMDCT-2048 + window, filter_bank.js:105-141,180-202 / mdct.js:62-115      */
/* ------------------------------------------------------------------------------------ */
/* xe[j] = X[2k], xo[j] = X[2k+1] for k = l + 64 j.  Writes the windowed second half to
 * tail[0..1023] (LDS, natural order) and returns the windowed first half at
 * n = 2 l + 128 m (hx[m]) and n + 1 (hy[m]).                                             */
DP_DEVICE void long_channel(const aacg_tables* T, int seq, int shape, int shape_prev, bool want_head,
                            const float (&xe)[8], const float (&xo)[8],
                            float* scratch, float* tail, float (&hx)[8], float (&hy)[8])
{
    const int l = dp_lane();

    /* X[N/2-1-2k] lives in lane 63-l as its odd value of element 7-j (mdct.js:74-75) */
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = xo[7 - j];
    dp_shfl(o, 63 - l);

    cpx z[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const aacg_c2 sc = T->sincos_long[l + 64 * j];
        z[j].im = dp_fma(xe[j], sc.re, o[j] * sc.im);                /* mdct.js:74 */
        z[j].re = dp_fma(o[j], sc.re, -(xe[j] * sc.im));             /* mdct.js:75 */
    }

    /* 512-point inverse FFT, unscaled (fft.js with forward = false) */
    radix8_inv(z);                                    /* over j (stride 64)       */
#pragma unroll
    for (int q = 1; q < 8; q++) z[q] = c_mul(z[q], T->tw512[q - 1][l]);
#pragma unroll
    for (int q = 0; q < 8; q++) lds_put(scratch, q * 72 + l, z[q]);
    dp_wave_sync();
    const int l0 = l & 7, qq = l >> 3;
#pragma unroll
    for (int j = 0; j < 8; j++) z[j] = lds_get(scratch, qq * 72 + l0 + 8 * j);
    dp_wave_sync();
    radix8_inv(z);                                    /* over l1 (stride 8)       */
#pragma unroll
    for (int r = 1; r < 8; r++) z[r] = c_mul(z[r], T->tw64[r - 1][l0]);
#pragma unroll
    for (int r = 0; r < 8; r++) lds_put(scratch, (qq + 8 * r) * 9 + l0, z[r]);
    dp_wave_sync();
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = lds_get(scratch, l * 9 + i);
    dp_wave_sync();
    radix8_inv(z);                                    /* over l0; lane l now holds Z[l + 64 r] */

    /* post-IFFT rotation (mdct.js:82-87), then fetch the mirror lane's values for the reorder */
    float m[16];
    const int lk = dp_opaque(l);                      /* same table entries as the pre-twiddle: re-read, not held */
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const aacg_c2 sc = T->sincos_long[lk + 64 * r];
        m[r]     = dp_fma(z[r].re, sc.re, -(z[r].im * sc.im));
        m[8 + r] = dp_fma(z[r].im, sc.re, z[r].re * sc.im);
    }
    float R[8], I[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { R[r] = m[r]; I[r] = m[8 + r]; }
    dp_shfl(m, 63 - l);                               /* m[r] = re[511 - k'], m[8+r] = im[..] of the mirror */

    /* reorder (mdct.js:90-114) fused with the window (filter_bank.js:109-116 etc.) */
    const float* hw = T->head_win[(seq == AACG_LONG_STOP_SEQUENCE ? 2 : 0) + shape_prev];
    const float* tw = T->tail_win[(seq == AACG_LONG_START_SEQUENCE ? 2 : 0) + shape];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = 2 * l + 128 * j;
        if (want_head) {
            dpf2 w0 = *(const dpf2*)(hw + n), w1 = *(const dpf2*)(hw + n + 512);
            hx[j]     = I[j + 4] * w0.x;              /* y[2k]        =  im[N/8 + k]     */
            hy[j]     = -m[3 - j] * w0.y;             /* y[2k+1]      = -re[N/8 - 1 - k] */
            hx[j + 4] = R[j] * w1.x;                  /* y[N/4+2k]    =  re[k]           */
            hy[j + 4] = -m[8 + 7 - j] * w1.y;         /* y[N/4+2k+1]  = -im[N/4 - 1 - k] */
        }
        dpf2 v0 = *(const dpf2*)(tw + n), v1 = *(const dpf2*)(tw + n + 512), t;
        t.x = R[j + 4] * v0.x;                        /* y[N/2+2k]    =  re[N/8 + k]     */
        t.y = -m[8 + 3 - j] * v0.y;                   /* y[N/2+2k+1]  = -im[N/8 - 1 - k] */
        *(dpf2*)(tail + n) = t;
        t.x = -I[j] * v1.x;                           /* y[3N/4+2k]   = -im[k]           */
        t.y = m[7 - j] * v1.y;                        /* y[3N/4+2k+1] =  re[N/4 - 1 - k] */
        *(dpf2*)(tail + n + 512) = t;
    }
}

/* ------------------------------------------------------------------------------------ */
/* EIGHT_SHORT_SEQUENCE: 8 x IMDCT-256 + window + inner overlap-add, filter_bank.js:143-178 */
/* ------------------------------------------------------------------------------------ */
/* Lane group w handles window w.  With s[p], p = 0..1151, the windowed sum of the eight
 * blocks placed at frame position 448 + p:   out[448+p] = ov[448+p] + s[p]  (p < 576),
 * new overlap[p-576] = s[p] (p >= 576), new overlap[576..1023] = 0.  Returns s at
 * p = 128 w + 2 g + 16 m (hx[m]) and p + 1 (hy[m]); only p < 576 is meaningful there.   */
DP_DEVICE void short_channel(const aacg_tables* T, int shape, int shape_prev, bool want_head,
                             const float (&xe)[8], const float (&xo)[8],
                             float* scratch, float* tail, float (&hx)[8], float (&hy)[8])
{
    const int l = dp_lane(), w = l >> 3, g = l & 7;
    (void)want_head;

    float o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = xo[7 - j];
    dp_shfl(o, l ^ 7);                                /* X_w[127 - 2k] from lane (w, 7-g) */

    cpx z[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const aacg_c2 sc = T->sincos_short[g + 8 * j];
        z[j].im = dp_fma(xe[j], sc.re, o[j] * sc.im);
        z[j].re = dp_fma(o[j], sc.re, -(xe[j] * sc.im));
    }

    /* 64-point inverse FFT per window: 8 lanes x 8 points */
    radix8_inv(z);
#pragma unroll
    for (int q = 1; q < 8; q++) z[q] = c_mul(z[q], T->tw64[q - 1][g]);
#pragma unroll
    for (int q = 0; q < 8; q++) lds_put(scratch, (8 * w + q) * 9 + g, z[q]);
    dp_wave_sync();
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = lds_get(scratch, l * 9 + i);
    dp_wave_sync();
    radix8_inv(z);                                    /* lane (w, q) holds Z_w[q + 8 r] */

    float m[16];
    const int gk = dp_opaque(g);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const aacg_c2 sc = T->sincos_short[gk + 8 * r];
        m[r]     = dp_fma(z[r].re, sc.re, -(z[r].im * sc.im));
        m[8 + r] = dp_fma(z[r].im, sc.re, z[r].re * sc.im);
    }
    float R[8], I[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { R[r] = m[r]; I[r] = m[8 + r]; }
    dp_shfl(m, l ^ 7);

    /* window each block: head with W[i] (block 0: previous shape), tail with W[127-i] */
    const float* ws = T->short_win[shape];
    const float* wh = (w == 0) ? T->short_win[shape_prev] : ws;
    float hd[16], tl[16];                             /* [m] = position i = 2g+16m, [8+m] = i+1 */
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = 2 * g + 16 * j;
        hd[j]         = I[j + 4] * wh[i];             /* y[2k]       */
        hd[8 + j]     = -m[3 - j] * wh[i + 1];        /* y[2k+1]     */
        hd[j + 4]     = R[j] * wh[i + 64];            /* y[64+2k]    */
        hd[8 + j + 4] = -m[8 + 7 - j] * wh[i + 65];   /* y[64+2k+1]  */
        tl[j]         = R[j + 4] * ws[127 - i];       /* y[128+2k]   */
        tl[8 + j]     = -m[8 + 3 - j] * ws[126 - i];  /* y[128+2k+1] */
        tl[j + 4]     = -I[j] * ws[63 - i];           /* y[192+2k]   */
        tl[8 + j + 4] = m[7 - j] * ws[62 - i];        /* y[192+2k+1] */
    }
    /* s[128 w + i] = tail of block w-1 + head of block w (filter_bank.js:155-160) */
    float pt[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pt[i] = tl[i];
    dp_shfl(pt, (l - 8) & 63);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        hx[i] = (w == 0 ? 0.0f : pt[i]) + hd[i];
        hy[i] = (w == 0 ? 0.0f : pt[8 + i]) + hd[8 + i];
    }

    /* second half of s -> new overlap (filter_bank.js:164-176) */
#pragma unroll
    for (int mm = 0; mm < 8; mm++) {
        const int p = 128 * w + 2 * g + 16 * mm;
        if (p >= 576) { dpf2 t; t.x = hx[mm]; t.y = hy[mm]; *(dpf2*)(tail + p - 576) = t; }
        if (w == 7)   { dpf2 t; t.x = tl[mm]; t.y = tl[8 + mm]; *(dpf2*)(tail + 448 + 2 * g + 16 * mm) = t; }
    }
#pragma unroll
    for (int t4 = 0; t4 < 4; t4++) {
        const int n = 576 + 2 * l + 128 * t4;
        if (n < 1024) { dpf2 zz; zz.x = 0.0f; zz.y = 0.0f; *(dpf2*)(tail + n) = zz; }
    }
}

/* ------------------------------------------------------------------------------------ */
/* spectral reconstruction: dequant (ics.js:222-227,244-256), MS (decoder.js:379-404),       */
/* IS (decoder.js:337-376)                                                                 */
/* ------------------------------------------------------------------------------------ */
struct chan_ctx {
    int cls;            /* 1 = EIGHT_SHORT lane map */
    int max_sfb;
    int group_count;
    int cum[7];         /* prefix sums of group_len: window w is in group #{i : w >= cum[i]} */
};

DP_DEVICE void load_ctx(const aacg_chan_info* ci, chan_ctx& cc)
{
    cc.cls = (ci->window_sequence == AACG_EIGHT_SHORT_SEQUENCE) ? 1 : 0;
    cc.max_sfb = ci->max_sfb;
    cc.group_count = ci->group_count;
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) { acc += ci->group_len[i]; cc.cum[i] = acc; }
}

/* index g*maxSFB + sfb of the band holding array position pos (ics.js:217), coded = sfb < maxSFB */
DP_DEVICE int band_index(const aacg_tables* T, const chan_ctx& cc, int pos, bool& coded)
{
    int sfb, g = 0;
    if (cc.cls) {
        sfb = T->band_of_short[pos & 127];
        const int w = pos >> 7;
#pragma unroll
        for (int i = 0; i < 7; i++) g += (i + 1 < cc.group_count && w >= cc.cum[i]) ? 1 : 0;
    } else {
        sfb = T->band_of_long[pos];
    }
    coded = sfb < cc.max_sfb;
    return coded ? g * cc.max_sfb + sfb : 0;
}

DP_DEVICE float meta_scale(const aacg_tables* T, unsigned mword)
{
    float sf = T->sf[mword & AACG_META_SF_MASK];
    return (mword & AACG_META_NEGATE) ? -sf : sf;
}

DP_DEVICE float dequant_one(const aacg_tables* T, int q, float sf)
{
    int a = q < 0 ? -q : q;
    a = a > 8191 ? 8191 : a;                          /* IQ_TABLE[8191..] is undefined in JS -> NaN */
    const float v = T->iq[a];
    return (q > 0 ? v : -v) * sf;                     /* q == 0 gives -0 like ics.js:251 */
}

DP_DEVICE void dequant_pair(const aacg_tables* T, unsigned mword, bool coded, int packed,
                            float& x0, float& x1)
{
    const int bt = (int)(mword >> AACG_META_BT_SHIFT);
    if (!coded || bt == AACG_ZERO_BT || bt >= AACG_NOISE_BT) {    /* ZERO / INTENSITY -> +0 (ics.js:222-227); NOISE: see DESIGN.md */
        x0 = 0.0f; x1 = 0.0f;
        return;
    }
    const float sf = meta_scale(T, mword);
    x0 = dequant_one(T, (int)(short)(packed & 0xffff), sf);
    x1 = dequant_one(T, packed >> 16, sf);
}

/* Fills (xe0, xo0) and, for a CPE, (xe1, xo1): the pair at pair_pos(lane, j, cls_c) of each
 * channel after dequant, MS and IS. */
DP_DEVICE void spectral_quant(const aacg_kparams& P, const aacg_unit_desc* u, int n_ch,
                              const chan_ctx& ccL, const chan_ctx& ccR, float* scratch,
                              float (&xe0)[8], float (&xo0)[8], float (&xe1)[8], float (&xo1)[8])
{
    const aacg_tables* T = P.tab;
    const int lane = dp_lane();
    const int16_t* q0 = (const int16_t*)P.coeffs + (size_t)u->coef_offset * 1024u;
    const aacg_band_meta* mL = P.meta + u->meta_offset;

    if (n_ch == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pos = pair_pos(lane, j, ccL.cls);
            bool coded; const int idx = band_index(T, ccL, pos, coded);
            dequant_pair(T, mL->band[idx], coded, *(const int*)(q0 + pos), xe0[j], xo0[j]);
        }
        return;
    }

    const aacg_band_meta* mR = mL + 1;
    const int16_t* q1 = q0 + 1024;
    const bool common = (u->flags & AACG_UNIT_COMMON_WINDOW) != 0;
    const bool mask   = (u->flags & AACG_UNIT_MASK_PRESENT) != 0;

    if (ccL.cls == ccR.cls) {
        /* both channels on the same lane map: MS and IS are register-local */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pos = pair_pos(lane, j, ccL.cls);
            bool codedL, codedR;
            const int idxL = band_index(T, ccL, pos, codedL);
            const int idxR = band_index(T, ccR, pos, codedR);
            const unsigned wL = mL->band[idxL], wR = mR->band[idxR];
            float a0, a1, b0, b1;
            dequant_pair(T, wL, codedL, *(const int*)(q0 + pos), a0, a1);
            dequant_pair(T, wR, codedR, *(const int*)(q1 + pos), b0, b1);
            /* decoder.js:295-296,393: MS needs commonWindow && maskPresent && ms_used && both band types < NOISE */
            if (common && mask && codedL && (wL & AACG_META_MS_USED) &&
                (wL >> AACG_META_BT_SHIFT) < AACG_NOISE_BT && (wR >> AACG_META_BT_SHIFT) < AACG_NOISE_BT) {
                const float t0 = a0 - b0, t1 = a1 - b1;
                a0 = a0 + b0; a1 = a1 + b1;
                b0 = t0; b1 = t1;
            }
            /* decoder.js:353-368: right = left * (c * sfR) on intensity bands of the right channel */
            const int btR = (int)(wR >> AACG_META_BT_SHIFT);
            if (codedR && btR >= AACG_INTENSITY_BT2) {
                float scale = meta_scale(T, wR);
                bool neg = (btR == AACG_INTENSITY_BT2);
                if (mask && (mL->band[idxR] & AACG_META_MS_USED)) neg = !neg;
                scale = neg ? -scale : scale;
                b0 = a0 * scale; b1 = a1 * scale;
            }
            xe0[j] = a0; xo0[j] = a1; xe1[j] = b0; xo1[j] = b1;
            if (j & 1) dp_sched_fence();
        }
        return;
    }

    /* L and R on different lane maps (no common window, one of them EIGHT_SHORT): MS cannot
     * apply (decoder.js:295); IS reads the left spectrum at the right channel's positions,
     * staged through LDS in natural order. */
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = pair_pos(lane, j, ccL.cls);
        bool coded; const int idx = band_index(T, ccL, pos, coded);
        dequant_pair(T, mL->band[idx], coded, *(const int*)(q0 + pos), xe0[j], xo0[j]);
        dpf2 t; t.x = xe0[j]; t.y = xo0[j];
        *(dpf2*)(scratch + pos) = t;
    }
    dp_wave_sync();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pos = pair_pos(lane, j, ccR.cls);
        bool coded; const int idx = band_index(T, ccR, pos, coded);
        const unsigned wR = mR->band[idx];
        dequant_pair(T, wR, coded, *(const int*)(q1 + pos), xe1[j], xo1[j]);
        const int btR = (int)(wR >> AACG_META_BT_SHIFT);
        if (coded && btR >= AACG_INTENSITY_BT2) {
            float scale = meta_scale(T, wR);
            bool neg = (btR == AACG_INTENSITY_BT2);
            if (mask && (mL->band[idx] & AACG_META_MS_USED)) neg = !neg;
            scale = neg ? -scale : scale;
            const dpf2 lv = *(const dpf2*)(scratch + pos);
            xe1[j] = lv.x * scale; xo1[j] = lv.y * scale;
        }
    }
    dp_wave_sync();
}

DP_DEVICE void load_f32(const float* x, int cls, float (&xe)[8], float (&xo)[8])
{
    const int lane = dp_lane();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const dpf2 v = *(const dpf2*)(x + pair_pos(lane, j, cls));
        xe[j] = v.x; xo[j] = v.y;
    }
}

/* IMDCT + window of one channel: tail -> LDS, windowed head -> registers */
DP_DEVICE void filter_channel(const aacg_tables* T, const aacg_chan_info* ci, int cls, bool want_head,
                              const float (&xe)[8], const float (&xo)[8],
                              float* scratch, float* tail, float (&hx)[8], float (&hy)[8])
{
    if (cls) short_channel(T, ci->window_shape, ci->window_shape_prev, want_head, xe, xo, scratch, tail, hx, hy);
    else     long_channel(T, ci->window_sequence, ci->window_shape, ci->window_shape_prev, want_head, xe, xo, scratch, tail, hx, hy);
}

/* out = (overlap + head) / 32768 for one channel, scalar stores at stride C (decoder.js:209-213) */
DP_DEVICE void store_channel(const float* pv, float* dst, int C, int cls, const float (&hx)[8], const float (&hy)[8])
{
    const int lane = dp_lane(), w = lane >> 3, g = lane & 7;
    const float S = 1.0f / 32768.0f;
    if (!cls) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int n = 2 * lane + 128 * m;
            const dpf2 a = *(const dpf2*)(pv + n);
            dst[(size_t)n * C]       = (a.x + hx[m]) * S;
            dst[(size_t)(n + 1) * C] = (a.y + hy[m]) * S;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            if (w < 4 || (w == 4 && m < 4)) {
                const int n = 448 + 128 * w + 2 * g + 16 * m;
                const dpf2 a = *(const dpf2*)(pv + n);
                dst[(size_t)n * C]       = (a.x + hx[m]) * S;
                dst[(size_t)(n + 1) * C] = (a.y + hy[m]) * S;
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; t4++) {               /* out[0..447] = overlap (filter_bank.js:149-151) */
            const int n = 2 * lane + 128 * t4;
            if (n < 448) {
                const dpf2 a = *(const dpf2*)(pv + n);
                dst[(size_t)n * C]       = a.x * S;
                dst[(size_t)(n + 1) * C] = a.y * S;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* one run per workgroup                                                                   */
/* ------------------------------------------------------------------------------------ */
template <int KIND>
DP_DEVICE void imdct_run_body(const aacg_kparams& P)
{
    const int lane = dp_lane(), wave = dp_wave();
    const aacg_run* run = P.runs + dp_block();
    float* lds = (float*)dp_lds();
    float* slot = lds + wave * AACG_SLOT_FLOATS;       /* tail[0] | tail[1] = FFT scratch */
    float* scratch = slot + 1024;

    const int n_units = run->n_units;
    int ui = -1;
    if (wave == 0) ui = run->pred_unit;
    else if (wave - 1 < n_units) ui = run->unit[wave - 1];
    ui = dp_uniform(ui);

    float hx0[8], hy0[8], hx1[8], hy1[8];
    const aacg_unit_desc* u = P.units;
    int n_ch = 0, cls0 = 0, cls1 = 0;

    if (ui >= 0) {
        u = P.units + ui;
        n_ch = u->n_ch;
        const bool want_head = wave != 0;
        /* TNS between here and the filterbank: identity as the reference runs (tns.js:106,122) */
        if (KIND == AACG_INPUT_QUANT_I16) {
            chan_ctx ccL, ccR;
            load_ctx(&u->ch[0], ccL);
            load_ctx(&u->ch[1], ccR);
            cls0 = ccL.cls; cls1 = ccR.cls;
            float xe0[8], xo0[8], xe1[8], xo1[8];
            spectral_quant(P, u, n_ch, ccL, ccR, scratch, xe0, xo0, xe1, xo1);
            filter_channel(P.tab, &u->ch[0], cls0, want_head, xe0, xo0, scratch, slot, hx0, hy0);
            if (n_ch == 2) {
                dp_wave_sync();                        /* tail[1] aliases the FFT scratch */
                filter_channel(P.tab, &u->ch[1], cls1, want_head, xe1, xo1, scratch, slot + 1024, hx1, hy1);
            }
        } else {
            const float* x = (const float*)P.coeffs + (size_t)u->coef_offset * 1024u;
            cls0 = u->ch[0].window_sequence == AACG_EIGHT_SHORT_SEQUENCE;
            cls1 = u->ch[1].window_sequence == AACG_EIGHT_SHORT_SEQUENCE;
            float xe0[8], xo0[8], xe1[8], xo1[8];
            load_f32(x, cls0, xe0, xo0);
            if (n_ch == 2) load_f32(x + 1024, cls1, xe1, xo1);
            filter_channel(P.tab, &u->ch[0], cls0, want_head, xe0, xo0, scratch, slot, hx0, hy0);
            if (n_ch == 2) {
                dp_wave_sync();
                filter_channel(P.tab, &u->ch[1], cls1, want_head, xe1, xo1, scratch, slot + 1024, hx1, hy1);
            }
        }
    } else if (wave == 0) {
        /* first run of its chain in this launch: the tail comes from the overlap state
         * (filter_bank.js:38-41, `overlap = this.overlaps[channel]`) */
        const aacg_unit_desc* u0 = P.units + run->unit[0];
        const int nc = u0->n_ch;
        for (int c = 0; c < nc; c++) {
            const float* src = P.overlap + (P.flip ? run->ov_b[c] : run->ov_a[c]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                *(dpf4*)(slot + c * 1024 + 4 * lane + 256 * i) = *(const dpf4*)(src + 4 * lane + 256 * i);
        }
    }

    dp_block_sync();

    if (wave >= 1 && ui >= 0) {
        const float* prev = lds + (wave - 1) * AACG_SLOT_FLOATS;
        const float S = 1.0f / 32768.0f;               /* decoder.js:211 */
        const int C = u->n_out_ch;
        float* pcm = P.pcm + u->pcm_offset + u->channel;
        const int w = lane >> 3, g = lane & 7;

        if (n_ch == 2 && C == 2 && cls0 == cls1 && ((u->pcm_offset | u->channel) & 3) == 0) {
            /* stereo fast path: (L[n], R[n], L[n+1], R[n+1]) = 16 bytes per lane */
            if (!cls0) {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int n = 2 * lane + 128 * m;
                    const dpf2 a = *(const dpf2*)(prev + n), b = *(const dpf2*)(prev + 1024 + n);
                    dpf4 o;
                    o.x = (a.x + hx0[m]) * S; o.y = (b.x + hx1[m]) * S;
                    o.z = (a.y + hy0[m]) * S; o.w = (b.y + hy1[m]) * S;
                    *(dpf4*)(pcm + 2 * n) = o;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    if (w < 4 || (w == 4 && m < 4)) {
                        const int n = 448 + 128 * w + 2 * g + 16 * m;
                        const dpf2 a = *(const dpf2*)(prev + n), b = *(const dpf2*)(prev + 1024 + n);
                        dpf4 o;
                        o.x = (a.x + hx0[m]) * S; o.y = (b.x + hx1[m]) * S;
                        o.z = (a.y + hy0[m]) * S; o.w = (b.y + hy1[m]) * S;
                        *(dpf4*)(pcm + 2 * n) = o;
                    }
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++) {       /* out[0..447] = overlap (filter_bank.js:149-151) */
                    const int n = 2 * lane + 128 * t4;
                    if (n < 448) {
                        const dpf2 a = *(const dpf2*)(prev + n), b = *(const dpf2*)(prev + 1024 + n);
                        dpf4 o; o.x = a.x * S; o.y = b.x * S; o.z = a.y * S; o.w = b.y * S;
                        *(dpf4*)(pcm + 2 * n) = o;
                    }
                }
            }
        } else {
            store_channel(prev, pcm, C, cls0, hx0, hy0);
            if (n_ch == 2) store_channel(prev + 1024, pcm + 1, C, cls1, hx1, hy1);
        }

        /* the chain's last frame in this launch: its tail is the new overlap state */
        if (wave == n_units && run->is_last) {
            for (int c = 0; c < n_ch; c++) {
                float* dstov = P.overlap + (P.flip ? run->ov_a[c] : run->ov_b[c]);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *(dpf4*)(dstov + 4 * lane + 256 * i) = *(const dpf4*)(slot + c * 1024 + 4 * lane + 256 * i);
            }
        }
    }
}

/* Spectral stage alone (one wave per unit): spec_out in ICStream.data order. */
DP_DEVICE void spectral_body(const aacg_kparams& P)
{
    const int lane = dp_lane();
    const aacg_unit_desc* u = P.units + dp_block();
    float* scratch = (float*)dp_lds();
    chan_ctx ccL, ccR;
    load_ctx(&u->ch[0], ccL);
    load_ctx(&u->ch[1], ccR);
    const int n_ch = u->n_ch;
    float xe0[8], xo0[8], xe1[8], xo1[8];
    spectral_quant(P, u, n_ch, ccL, ccR, scratch, xe0, xo0, xe1, xo1);
    float* out = P.spec_out + (size_t)u->coef_offset * 1024u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        dpf2 t; t.x = xe0[j]; t.y = xo0[j];
        *(dpf2*)(out + pair_pos(lane, j, ccL.cls)) = t;
    }
    if (n_ch == 2) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            dpf2 t; t.x = xe1[j]; t.y = xo1[j];
            *(dpf2*)(out + 1024 + pair_pos(lane, j, ccR.cls)) = t;
        }
    }
}

#endif /* AACG_KERNELS_H */
