/*
 * aacg_kernels8.h — the run kernels at 8 waves per SIMD: one CHANNEL per wave.
 *
 * The 16-wave kernels of aacg_kernels.h carry a channel pair per wave: 120 VGPRs and an 8 KB LDS slot, which pins a CU at
 * sixteen waves (four per SIMD), and four instruction streams per SIMD cover only part of each other's latencies
 * (DESIGN.md 6b).  Here a wave carries ONE channel of one frame — a 4 KB slot, at most 64 VGPRs — and two 16-wave
 * workgroups share a CU.  What had to change for that (reference stages: fft.js:140-191, mdct.js:73-114,
 * filter_bank.js:105-178, decoder.js:203-215, 337-404):
 *
 *   arithmetic   one complex value per register pair, (re, im) packed: a butterfly, a rotation or a twiddle is one or two
 *                v_pk_*_f32 whichever channel it belongs to (the 16-wave kernels pack (left, right) instead).
 *   dequant      the two waves of a channel pair split the SPECTRUM, not the channels: wave h dequantises coefficients
 *                512 h .. 512 h + 511 of both channels, so M/S and intensity stay register-local (decoder.js:337-404), and
 *                writes each channel's values into that channel's slot; one flag exchange, then wave c transforms channel c.
 *   transform    N/4-point complex inverse FFT between the two rotations of mdct.js:73-87 as radix-8 register butterflies
 *                with two LDS transposes (long) or one (eight short windows side by side), as before.
 *   epilogue     the wave does NOT window: it leaves the rotated FFT output (re, im) of its channel in its slot, and the
 *                reorder of mdct.js:90-114, the windows of filter_bank.js:105-178, the overlap-add and the interleave of
 *                decoder.js:203-215 happen in one pass that reads the slots of the frame and of the frame before it at whatever
 *                addresses the output order needs — no mirror-lane exchange, no values held in registers across the
 *                wait for the previous frame.  The two waves of a pair split the SAMPLES (wave h writes samples 512 h ..
 *                of both channels as 16-byte (L, R, L, R) stores).
 *   windows      read from global memory / L2 in output order (aacg_win8: START / STOP shapes composed on the host); two
 *                copies of the old 25 KB table block would not fit beside 2 x 64 KB of slots.
 *   chains       a run is 8 frames of a pair or 16 of a single channel; consecutive runs of a chain hand the windowed tail
 *                over through global memory by a rendezvous: the first of the two sides to arrive publishes what it has
 *                (the tail, or the windowed first half of the next frame) and leaves, the second one finishes the frame.
 *                Nothing waits for another workgroup, so no dispatch order is assumed and no IMDCT is recomputed.
 *
 * The overlap-add is tail + head with both terms rounded products (never a fused multiply-add): whichever side finishes a
 * frame, and however a batch is cut, the same bits come out.
 */
#ifndef AACG_KERNELS8_H
#define AACG_KERNELS8_H

#include "aacg_kernels.h"

#ifdef AACG_PROFILE
#define K8_TRACE(k) do { if (trace && lane == 0) trace[k] = dp_clock(); } while (0)
#else
#define K8_TRACE(k) do { } while (0)
#endif

#ifndef AACG8_EARLY_WAVES
#define AACG8_EARLY_WAVES 8
#endif
#define K8_PHASE_STAGED 1
#define K8_PHASE_DUMPED 2

/* `tabq + AACG_TAB_OFF_SF` etc. address this kernel's LDS table block, which has no windows in it */
#define K8_TABQ(tab) ((tab) - AACG8_WIN_GAP_FLOATS)

/* staging swizzle of the E / O planes (pair index k of X[2k] at k, of X[2k+1] at 512 + k): keeps the long (l + 64 j) and
 * the short (64 w + g + 8 j) read patterns and the 16-byte staging stores conflict-free */
DP_DEVICE int k8_stg(int k) { return k ^ (((k >> 6) & 7) << 3); }
/* where a short window's rotated output k (0..63) of window w sits in the R / I planes of the slot */
DP_DEVICE int k8_sdump(int w, int k) { return 64 * w + (k ^ ((w & 7) << 3)); }

/* Long window: staged spectrum in the slot -> the rotated FFT output (R, I)[l + 64 r] in z[r] (mdct.js:73-87 around
 * fft.js:140-191), then left in the slot as planar R[512], I[512].  The slot is reused for the two transposes.  Every LDS
 * address is a lane-dependent base, an XOR with a constant and an immediate offset (slots start on 512-byte boundaries):
 * built from the lane number element by element the index arithmetic was 170 vector instructions per frame. */
DP_DEVICE void k8_long(const float* tab, float* slot)
{
    const int l = dp_lane();
    const int sb = dp_lds_addr(slot);
    const int tb = dp_lds_addr(tab + AACG_TAB_OFF_SINCOS_LONG) + 8 * l;
    dpv2 z[8];
    {
        /* E[stg(k)], k = l + 64 j, at 256 j + (4 l ^ 32 j);  O[stg(511 - k)] at 2048 + 256 (7 - j) + (4 (63 - l) ^ 32 (7 - j)) */
        const int a0 = sb + 4 * l, b0 = sb + 2048 + 4 * (63 - l);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const dpv2 sc = dp_lds_read_v2(tb + 512 * j);
            const float a = dp_lds_read_f32((a0 ^ (32 * j)) + 256 * j);                   /* X[2k]        */
            const float b = dp_lds_read_f32((b0 ^ (32 * (7 - j))) + 256 * (7 - j));       /* X[1023 - 2k] */
            z[j] = dp_cmul(k8_v2(b, a), sc);                                              /* mdct.js:74-75 */
        }
    }
    dp_wave_sync();
    k8_radix8(z);
    {
        const int t5 = dp_lds_addr(tab + AACG_TAB_OFF_TW512) + 8 * l;
#pragma unroll
        for (int q = 1; q < 8; q++) z[q] = dp_cmul(z[q], dp_lds_read_v2(t5 + 512 * (q - 1)));
        const int w0 = sb + 8 * l;                     /* xch1(q, l): 512 q + (8 l ^ 64 q) */
#pragma unroll
        for (int q = 0; q < 8; q++) dp_lds_write_v2((w0 ^ (64 * q)) + 512 * q, z[q]);
    }
    dp_wave_sync();
    const int l0 = l & 7, qq = l >> 3;
    {
        const int r0 = sb + 512 * qq + 8 * l0 + 64 * qq;      /* xch1(qq, l0 + 8 j): 512 qq + 8 l0 + 64 (j ^ qq) */
#pragma unroll
        for (int j = 0; j < 8; j++) z[j] = dp_lds_read_v2(r0 ^ (64 * j));
    }
    dp_wave_sync();
    k8_radix8(z);
    {
        const int t6 = dp_lds_addr(tab + AACG_TAB_OFF_TW64) + 8 * l0;
#pragma unroll
        for (int r = 1; r < 8; r++) z[r] = dp_cmul(z[r], dp_lds_read_v2(t6 + 64 * (r - 1)));
        /* xch2(qq + 8 r, l0): 64 qq + 512 r + 8 (l0 ^ (qq >> 2) ^ (2 r & 6)) */
        const int x0 = sb + 64 * qq + 8 * (l0 ^ (qq >> 2));
#pragma unroll
        for (int r = 0; r < 8; r++) dp_lds_write_v2((x0 ^ ((16 * r) & 48)) + 512 * r, z[r]);
    }
    dp_wave_sync();
    {
        const int y0 = sb + 64 * l + 8 * ((l >> 2) & 7);      /* xch2(l, i): 64 l + 8 (i ^ (l >> 2 & 7)) */
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = dp_lds_read_v2(y0 ^ (8 * i));
    }
    dp_wave_sync();
    k8_radix8(z);                                      /* the lane holds Z[l + 64 r] */
    {
        const int d0 = sb + 4 * l;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const dpv2 ri = dp_cmul(z[r], dp_lds_read_v2(tb + 512 * r));                  /* mdct.js:82-87 */
            dp_lds_write_f32(d0 + 256 * r, ri[0]);
            dp_lds_write_f32(d0 + 2048 + 256 * r, ri[1]);
        }
    }
}

/* EIGHT_SHORT: eight 64-point transforms side by side, lane (w = l >> 3, g = l & 7); the rotated output (R, I)_w[g + 8 r]
 * goes to k8_sdump(w, g + 8 r) of the slot's R / I planes */
DP_DEVICE void k8_short(const float* tab, float* slot)
{
    const int l = dp_lane(), w = l >> 3, g = l & 7;
    const int sb = dp_lds_addr(slot);
    const int tb = dp_lds_addr(tab + AACG_TAB_OFF_SINCOS_SHORT) + 8 * g;
    const int t6 = dp_lds_addr(tab + AACG_TAB_OFF_TW64) + 8 * g;
    const int a0 = sb + 256 * w + 4 * g + 32 * w;             /* E[stg(64 w + g + 8 j)] = 256 w + 4 g + 32 (j ^ w) */
    dpv2 z[8];
    {
        const int b0 = sb + 2048 + 256 * w + 4 * (7 - g) + 32 * w;   /* O[stg(64 w + 63 - g - 8 j)] = 2048 + 256 w + 4 (7 - g) + 32 ((7 - j) ^ w) */
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const dpv2 sc = dp_lds_read_v2(tb + 64 * j);
            const float a = dp_lds_read_f32(a0 ^ (32 * j));                              /* X_w[2k], k = g + 8 j */
            const float b = dp_lds_read_f32(b0 ^ (32 * (7 - j)));                        /* X_w[127 - 2k]        */
            z[j] = dp_cmul(k8_v2(b, a), sc);
        }
    }
    dp_wave_sync();
    k8_radix8(z);
#pragma unroll
    for (int q = 1; q < 8; q++) z[q] = dp_cmul(z[q], dp_lds_read_v2(t6 + 64 * (q - 1)));
    {
        /* xch2(8 w + q, g): 512 w + 64 q + 8 (g ^ (2 w & 6) ^ (q >> 2)) */
        const int x0 = sb + 512 * w + 8 * (g ^ ((2 * w) & 6));
#pragma unroll
        for (int q = 0; q < 8; q++) dp_lds_write_v2((x0 ^ (8 * (q >> 2))) + 64 * q, z[q]);
    }
    dp_wave_sync();
    {
        const int y0 = sb + 64 * l + 8 * ((l >> 2) & 7);
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = dp_lds_read_v2(y0 ^ (8 * i));
    }
    dp_wave_sync();
    k8_radix8(z);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const dpv2 ri = dp_cmul(z[r], dp_lds_read_v2(tb + 64 * r));
        dp_lds_write_f32(a0 ^ (32 * r), ri[0]);                /* k8_sdump(w, g + 8 r) = 64 w + g + 8 (r ^ w) */
        dp_lds_write_f32((a0 ^ (32 * r)) + 2048, ri[1]);
    }
}

/* IMDCT output m (0..2047) of a long frame from its dump: the reorder of mdct.js:90-114,
 *   y[2k] = I[256+k]   y[2k+1] = -R[255-k]   y[512+2k] = R[k]   y[512+2k+1] = -I[511-k]
 *   y[1024+2k] = R[256+k]   y[1024+2k+1] = -I[255-k]   y[1536+2k] = -I[k]   y[1536+2k+1] = R[511-k]        (k < 256) */
DP_DEVICE float k8_y_long(const float* slot, int m)
{
    const int pc = m >> 9, kk = (m & 511) >> 1, par = m & 1;
    const int use_i = (((pc == 0) | (pc == 3)) ? 1 : 0) ^ par;
    const int idx = par ? ((pc & 1) ? 511 - kk : 255 - kk) : ((pc & 1) ? kk : 256 + kk);
    const float v = slot[(use_i ? 512 : 0) + idx];
    const bool neg = par ? (pc != 3) : (pc == 3);
    return neg ? -v : v;
}
/* output i (0..255) of short window w: the same reorder at N = 256 */
DP_DEVICE float k8_y_short(const float* slot, int w, int i)
{
    const int pc = i >> 6, kk = (i & 63) >> 1, par = i & 1;
    const int use_i = (((pc == 0) | (pc == 3)) ? 1 : 0) ^ par;
    const int idx = par ? ((pc & 1) ? 63 - kk : 31 - kk) : ((pc & 1) ? kk : 32 + kk);
    const float v = slot[(use_i ? 512 : 0) + k8_sdump(w, idx)];
    const bool neg = par ? (pc != 3) : (pc == 3);
    return neg ? -v : v;
}

struct k8_chan { int seq, shape, shape_prev; };

/* T(n): sample n of the windowed second half of a dumped frame — what FilterBank.overlaps holds after it
 * (filter_bank.js:114-116,129-139,164-176), PCM-scaled.  Every product is rounded on its own. */
DP_DEVICE float k8_tail(const aacg_win8* W, const float* slot, const k8_chan& c, int n)
{
#pragma clang fp contract(off)
    if (c.seq != AACG_EIGHT_SHORT_SEQUENCE) {
        const int tv = 2 * (c.seq == AACG_LONG_START_SEQUENCE ? 1 : 0) + c.shape;
        const float w = W->tail[tv][n];
        float t = k8_y_long(slot, 1024 + n) * w;
        if (tv >= 2 && w == 0.0f) t = 0.0f;            /* a region the window takes nothing from: NaN / Inf must not leak (filter_bank.js:129-139) */
        return t;
    }
    /* eight short windows: sample 1024 + n of s[448 + 128 w + i] += y_w[i] W[i] (filter_bank.js:153-176) */
    const int p = n + 576, wa = p >> 7, i = p & 127;
    float t = 0.0f;
    if (n < 576) {
        const float tl = k8_y_short(slot, wa - 1, 128 + i) * W->shrt[c.shape][127 - i];
        t = tl;
        if (wa <= 7) t = tl + k8_y_short(slot, wa, i) * W->shrt[c.shape][i];
    }
    return t;
}

/* H(n): sample n of the windowed first half of a dumped frame (filter_bank.js:109-111,124-126,153-160,185-195); -0.0 where the
 * sequence takes the sample from the overlap alone (EIGHT_SHORT: 0..447, filter_bank.js:149-151), so that T + H = T there */
DP_DEVICE float k8_head(const aacg_win8* W, const float* slot, const k8_chan& c, int n)
{
#pragma clang fp contract(off)
    if (c.seq != AACG_EIGHT_SHORT_SEQUENCE) {
        const int hv = 2 * (c.seq == AACG_LONG_STOP_SEQUENCE ? 1 : 0) + c.shape_prev;
        const float w = W->head[hv][n];
        float h = k8_y_long(slot, n) * w;
        if (hv >= 2 && w == 0.0f) h = 0.0f;
        return h;
    }
    float h = -0.0f;
    if (n >= 448) {
        const int wb = (n - 448) >> 7, i = (n - 448) & 127;
        const float hd = k8_y_short(slot, wb, i) * W->shrt[wb == 0 ? c.shape_prev : c.shape][i];
        const float pt = wb == 0 ? 0.0f : k8_y_short(slot, wb - 1, 128 + i) * W->shrt[c.shape][127 - i];
        h = pt + hd;
    }
    return h;
}

/* where the two terms of a frame's output come from: a dump in LDS (with the frame's window fields), or finished values
 * in global memory (the overlap state, or a rendezvous payload) */
struct k8_src { const float* lds[2]; k8_chan ch[2]; const float* glob[2]; };

DP_DEVICE float k8_eval_tail(const aacg_win8* W, const k8_src& s, int c, int n)
{
    if (s.glob[c]) return dp_g_load_f1(s.glob[c] + n);
    return k8_tail(W, s.lds[c], s.ch[c], n);
}
DP_DEVICE float k8_eval_head(const aacg_win8* W, const k8_src& s, int c, int n)
{
    if (s.glob[c]) return dp_g_load_f1(s.glob[c] + n);
    return k8_head(W, s.lds[c], s.ch[c], n);
}

/* The general finishing pass: samples n0 + lane + 64 t (t < NT) of NC channels, out = T + H, consecutive samples in
 * consecutive lanes (an element of a wider frame owns 1-2 of C interleaved channels: 64 sample-frames per store). */
template <int NC, int NT>
DP_DEVICE void k8_finish_general(const aacg_win8* W, const k8_src& T, const k8_src& H, float* pcm /* element's first channel, sample 0 */,
                                 int C, int n0)
{
#pragma clang fp contract(off)
    const int lane = dp_lane();
#pragma unroll 2
    for (int t = 0; t < NT; t++) {
        const int n = n0 + lane + 64 * t;
        const float a = k8_eval_tail(W, T, 0, n) + k8_eval_head(W, H, 0, n);
        if (NC == 2) {
            const float b = k8_eval_tail(W, T, 1, n) + k8_eval_head(W, H, 1, n);
            pcm_put2(pcm + (size_t)n * C, a, b);
        } else {
            pcm_put1(pcm + (size_t)n * C, a);
        }
    }
}

/* T (TAIL = true) or H of NC channels, sample pairs n0 + 2 lane + 128 t (t < NT), to planar arrays in global memory:
 * the chain's new overlap state (plain stores), or a rendezvous payload (PUBLISH: agent-scope write-through stores) */
template <int NC, int NT, bool TAIL, bool PUBLISH>
DP_DEVICE void k8_export(const aacg_win8* W, const k8_src& S, float* d0, float* d1, int n0)
{
    const int lane = dp_lane();
#pragma unroll 2
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 2 * lane + 128 * t;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float a = TAIL ? k8_eval_tail(W, S, c, n) : k8_eval_head(W, S, c, n);
            const float b = TAIL ? k8_eval_tail(W, S, c, n + 1) : k8_eval_head(W, S, c, n + 1);
            float* d = (c ? d1 : d0) + n;
            if (PUBLISH) dp_g_store_f2(d, a, b);
            else { dpf2 v; v.x = a; v.y = b; *(dpf2*)d = v; }
        }
    }
}

/* Sample pair (n, n + 1), n = 512 h + 2 k (k < 256), of a long-type frame's dump at byte address `base` (planar R at 0,
 * I at 2048): where its two IMDCT outputs sit, and their signs (the reorder of mdct.js:90-114):
 *   first half   h = 0: ( I[256+k], -R[255-k] )    h = 1: ( R[k],  -I[511-k] )
 *   second half  h = 0: ( R[256+k], -I[255-k] )    h = 1: ( -I[k],  R[511-k] )
 * The second element always runs downwards in k; `lo` is the address for k = lane + 64 j, j = 0, the others follow at
 * + 256 j (first element) and - 256 j (second element). */
struct k8_pair_addr { int first, second; };
DP_DEVICE k8_pair_addr k8_head_addr(int base, int h, int lane)
{
    k8_pair_addr a;
    a.first = base + 4 * lane + (h ? 0 : 3072);                /* R[k] | I[256 + k]     */
    a.second = base - 4 * lane + (h ? 4 * 1023 : 4 * 255);     /* I[511 - k] | R[255 - k] */
    return a;
}
DP_DEVICE k8_pair_addr k8_tail_addr(int base, int h, int lane)
{
    k8_pair_addr a;
    a.first = base + 4 * lane + (h ? 2048 : 1024);             /* I[k] | R[256 + k]     */
    a.second = base - 4 * lane + (h ? 4 * 511 : 4 * 767);      /* R[511 - k] | I[255 - k] */
    return a;
}

/* The stereo fast path: both channels of the frame and of the frame before it long-type with the same window tables
 * (ONLY_LONG / LONG_START / LONG_STOP: the shapes are tables), a frame of two interleaved channels.  Wave h finishes samples
 * 512 h + 2 k, + 1 for k = lane + 64 j: per j four (L, R) pairs from LDS (the two slots of a channel pair are 4 KB apart: one
 * two-address read each), two 8-byte window reads, six packed operations, one 16-byte (L[n], R[n], L[n+1], R[n+1]) store. */
struct k8_wins { dpf2 wh[4], wt[4]; };
/* the fast path's window values, requested BEFORE the wave waits for the previous frame (they depend on window fields only:
 * inside the pass the loads were a memory round trip on every wave's critical path) */
DP_DEVICE void k8_stereo_windows(const aacg_win8* W, const k8_src& T, const k8_src& H, int h, k8_wins& w)
{
    const int lane = dp_lane();
    const int hv = 2 * (H.ch[0].seq == AACG_LONG_STOP_SEQUENCE ? 1 : 0) + H.ch[0].shape_prev;
    const int tv = 2 * (T.ch[0].seq == AACG_LONG_START_SEQUENCE ? 1 : 0) + T.ch[0].shape;
    const float* whp = W->head[hv] + 512 * h + 2 * lane;
    const float* wtp = W->tail[tv] + 512 * h + 2 * lane;
#pragma unroll
    for (int j = 0; j < 4; j++) { w.wh[j] = *(const dpf2*)(whp + 128 * j); w.wt[j] = *(const dpf2*)(wtp + 128 * j); }
}
DP_DEVICE void k8_finish_stereo_long(const k8_wins& w, const k8_src& T, const k8_src& H, float* pcm, int h)
{
#pragma clang fp contract(off)
    const int lane = dp_lane();
    const int hv = 2 * (H.ch[0].seq == AACG_LONG_STOP_SEQUENCE ? 1 : 0) + H.ch[0].shape_prev;
    const int tv = 2 * (T.ch[0].seq == AACG_LONG_START_SEQUENCE ? 1 : 0) + T.ch[0].shape;
    const bool from_lds = T.glob[0] == nullptr, head_lds = H.glob[0] == nullptr;
    const bool guard = (head_lds && hv >= 2) || (from_lds && tv >= 2);       /* START / STOP shapes: regions the window takes nothing from */
    const k8_pair_addr ca = k8_head_addr(head_lds ? dp_lds_addr(H.lds[0]) : 0, h, lane);
    const k8_pair_addr pa = k8_tail_addr(from_lds ? dp_lds_addr(T.lds[0]) : 0, h, lane);
    /* the signs of the second half's two elements: (+, -) for h = 0, (-, +) for h = 1 */
    const unsigned s_first = h ? 0x80000000u : 0u, s_second = h ? 0u : 0x80000000u;
    float* out = pcm + 2 * (512 * h + 2 * lane);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        dpv2 h0, h1, t0, t1;                           /* (left, right) */
        dpf2 wh = w.wh[j];
        if (head_lds) {
            h0[0] = dp_lds_read_f32(ca.first + 256 * j);  h0[1] = dp_lds_read_f32(ca.first + 256 * j + 4096);
            h1[0] = dp_lds_read_f32(ca.second - 256 * j); h1[1] = dp_lds_read_f32(ca.second - 256 * j + 4096);
            h0 = h0 * k8_v2(wh.x, wh.x);
            h1 = h1 * k8_v2(-wh.y, -wh.y);
        } else {
            /* the next run's first frame, whose windowed first half that run left behind (rendezvous payload) */
            const int n = 512 * h + 2 * lane + 128 * j;
            const dpf2 hl = dp_g_load_f2(H.glob[0] + n), hr = dp_g_load_f2(H.glob[1] + n);
            h0 = k8_v2(hl.x, hr.x); h1 = k8_v2(hl.y, hr.y);
            wh.x = wh.y = 1.0f;
        }
        if (from_lds) {
            t0[0] = dp_lds_read_f32(pa.first + 256 * j);  t0[1] = dp_lds_read_f32(pa.first + 256 * j + 4096);
            t1[0] = dp_lds_read_f32(pa.second - 256 * j); t1[1] = dp_lds_read_f32(pa.second - 256 * j + 4096);
            const dpf2 wt = w.wt[j];
            const float wx = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, wt.x) ^ s_first);
            const float wy = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, wt.y) ^ s_second);
            t0 = t0 * k8_v2(wx, wx);
            t1 = t1 * k8_v2(wy, wy);
            if (guard) {
                dp_keep_branch();
                if (wt.x == 0.0f) t0 = k8_v2(0.0f, 0.0f);
                if (wt.y == 0.0f) t1 = k8_v2(0.0f, 0.0f);
            }
        } else {
            const int n = 512 * h + 2 * lane + 128 * j;
            const dpf2 tl = dp_g_load_f2(T.glob[0] + n), tr = dp_g_load_f2(T.glob[1] + n);
            t0 = k8_v2(tl.x, tr.x); t1 = k8_v2(tl.y, tr.y);
        }
        if (guard) {
            dp_keep_branch();
            if (wh.x == 0.0f) h0 = k8_v2(0.0f, 0.0f);
            if (wh.y == 0.0f) h1 = k8_v2(0.0f, 0.0f);
        }
        const dpv2 o0 = t0 + h0, o1 = t1 + h1;
        dpf4 o; o.x = o0[0]; o.y = o0[1]; o.z = o1[0]; o.w = o1[1];
        dp_store_nt((dpf4*)(out + 256 * j), o);
    }
}

/* T (TAIL) or H of one long-type channel, sample pairs 512 h + 2 lane + 128 j, + 1 (j < 4), to a planar array in global
 * memory: the chain's new overlap state (plain stores) or a rendezvous payload (PUBLISH: agent-scope write-through stores).
 * The same products, rounded the same way, as k8_tail / k8_head and the fast path form them. */
template <bool TAIL, bool PUBLISH>
DP_DEVICE void k8_export_long(const aacg_win8* W, const float* slot, const k8_chan& c, float* dst, int h)
{
#pragma clang fp contract(off)
    const int lane = dp_lane();
    const int v = TAIL ? 2 * (c.seq == AACG_LONG_START_SEQUENCE ? 1 : 0) + c.shape : 2 * (c.seq == AACG_LONG_STOP_SEQUENCE ? 1 : 0) + c.shape_prev;
    const float* wp = (TAIL ? W->tail[v] : W->head[v]) + 512 * h + 2 * lane;
    const k8_pair_addr a = TAIL ? k8_tail_addr(dp_lds_addr(slot), h, lane) : k8_head_addr(dp_lds_addr(slot), h, lane);
    const unsigned s_first = (TAIL && h) ? 0x80000000u : 0u, s_second = (TAIL && h) ? 0u : 0x80000000u;
    /* all window reads first: one memory round trip, not one per store */
    dpf2 w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = *(const dpf2*)(wp + 128 * j);
    float x0[4], x1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float y0 = dp_lds_read_f32(a.first + 256 * j), y1 = dp_lds_read_f32(a.second - 256 * j);
        x0[j] = y0 * __builtin_bit_cast(float, __builtin_bit_cast(unsigned, w[j].x) ^ s_first);
        x1[j] = y1 * __builtin_bit_cast(float, __builtin_bit_cast(unsigned, w[j].y) ^ s_second);
    }
    if (v >= 2) {
        dp_keep_branch();
#pragma unroll
        for (int j = 0; j < 4; j++) { if (w[j].x == 0.0f) x0[j] = 0.0f; if (w[j].y == 0.0f) x1[j] = 0.0f; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float* d = dst + 512 * h + 2 * lane + 128 * j;
        if (PUBLISH) dp_g_store_f2(d, x0[j], x1[j]);
        else { dpf2 o; o.x = x0[j]; o.y = x1[j]; *(dpf2*)d = o; }
    }
}

/* T or H of `nc` channels to planar arrays: samples 512 h .. 512 h + 511, or (whole) all 1024; long-type channels through the
 * table-driven pass above, EIGHT_SHORT channels through the general evaluation */
template <bool TAIL, bool PUBLISH>
DP_DEVICE void k8_export_any(const aacg_win8* W, const k8_src& S, int nc, bool whole, int h, float* d0, float* d1)
{
#pragma unroll
    for (int c = 0; c < 2; c++) {
        if (c < nc) {
            float* dst = c ? d1 : d0;
            if (S.ch[c].seq != AACG_EIGHT_SHORT_SEQUENCE) {
                if (whole) { k8_export_long<TAIL, PUBLISH>(W, S.lds[c], S.ch[c], dst, 0); k8_export_long<TAIL, PUBLISH>(W, S.lds[c], S.ch[c], dst, 1); }
                else k8_export_long<TAIL, PUBLISH>(W, S.lds[c], S.ch[c], dst, h);
            } else {
                dp_keep_branch();
                k8_src one;
                one.lds[0] = one.lds[1] = S.lds[c]; one.ch[0] = one.ch[1] = S.ch[c]; one.glob[0] = one.glob[1] = nullptr;
                if (whole) k8_export<1, 8, TAIL, PUBLISH>(W, one, dst, dst, 0);
                else       k8_export<1, 4, TAIL, PUBLISH>(W, one, dst, dst, 512 * h);
            }
        }
    }
}

/* ---- dequantisation (ics.js:222-227,244-256), M/S (decoder.js:379-404), intensity (decoder.js:337-376) ------------------ */
/* band records as in aacg_kernels.h (prepare_bands), left and right at separate bases */
template <int H0, int H1, bool TWO>
DP_DEVICE void k8_prepare_bands(const unsigned (&mw)[2][2], const float* tabq, bool ms_on, bool mask, float* btl, float* btr)
{
    const int lane = dp_lane();
    const unsigned kCoded = 0x1FFEu, kBelowNoise = 0x1FFFu;
    const unsigned kLiveR = TWO ? kCoded : 0u, kMsL = ms_on ? kBelowNoise : 0u;
    const unsigned kIs = TWO ? 0xC000u : 0u;
    const unsigned mask_u = mask ? 1u : 0u;
    float sl_in[2], sr_in[2];
#pragma unroll
    for (int h = H0; h < H1; h++) {
        sl_in[h] = tabq[AACG_TAB_OFF_SF + (mw[0][h] & AACG_META_SF_MASK)];
        sr_in[h] = TWO ? tabq[AACG_TAB_OFF_SF + (mw[1][h] & AACG_META_SF_MASK)] : 0.0f;
    }
#pragma unroll
    for (int h = H0; h < H1; h++) {
        const int b = lane + 64 * h;
        const bool coded = b < AACG_MAX_SECTIONS;
        const unsigned wl = coded ? mw[0][h] : 0u, wr = (TWO && coded) ? mw[1][h] : 0u;
        const unsigned tl = wl >> AACG_META_BT_SHIFT, tr = wr >> AACG_META_BT_SHIFT;
        const unsigned live_l = 0u - ((kCoded >> tl) & 1u), live_r = 0u - ((kLiveR >> tr) & 1u);
        const unsigned ms_used = (wl / AACG_META_MS_USED) & 1u;
        const unsigned ms = ms_used & (kMsL >> tl) & (kBelowNoise >> tr) & 1u;
        const unsigned is = (kIs >> tr) & 1u;
        const unsigned neg = ((0x4000u >> tr) ^ (mask_u & ms_used)) & 1u;
        const unsigned sl = __builtin_bit_cast(unsigned, sl_in[h]) ^ ((wl * (0x80000000u / AACG_META_NEGATE)) & 0x80000000u);
        const unsigned sr = __builtin_bit_cast(unsigned, sr_in[h]) ^ ((wr * (0x80000000u / AACG_META_NEGATE)) & 0x80000000u);
        dpf2 rl, rr;
        rl.x = __builtin_bit_cast(float, sl & live_l);
        rl.y = __builtin_bit_cast(float, (live_l & AACG_BR_LIVE) | ms);
        *(dpf2*)(btl + 2 * b) = rl;
        if (TWO) {
            rr.x = __builtin_bit_cast(float, (sr & live_r) | ((sr ^ (neg << 31)) & (0u - is)));
            rr.y = __builtin_bit_cast(float, (live_r & AACG_BR_LIVE) | is);
            *(dpf2*)(btr + 2 * b) = rr;
        }
    }
}

/* band record indices of the lane's two 4-coefficient groups at 8 lane + 512 i + {0, 4} */
DP_DEVICE void k8_band_idx(const float* tabq, const chan_ctx& cc, int i, int (&idx)[2])
{
    const int lane = dp_lane();
    const unsigned char* bl = (const unsigned char*)(tabq + AACG_TAB_OFF_BAND_LONG);
    const unsigned char* bs = (const unsigned char*)(tabq + AACG_TAB_OFF_BAND_SHORT);
    const dpf2 ml = *(const dpf2*)(bl + 8 * lane + 512 * i);
    const dpf2 ms = *(const dpf2*)(bs + ((8 * lane) & 127));
    const dpf2 m = cc.cls ? ms : ml;
    const int s0 = (int)(__builtin_bit_cast(unsigned, m.x) & 0xffu), s1 = (int)(__builtin_bit_cast(unsigned, m.y) & 0xffu);
    const int g = cc.cls ? (int)((cc.gmap >> (4 * ((lane >> 4) + 4 * i))) & 15u) : 0;
    idx[0] = s0 < cc.max_sfb ? g * cc.max_sfb + s0 : AACG_BR_NONE;
    idx[1] = s1 < cc.max_sfb ? g * cc.max_sfb + s1 : AACG_BR_NONE;
}

/* coefficients 8 lane + 512 i + 0..7 of the left (single) channel and, TWO, of the right: dequantised, M/S and intensity applied */
template <bool TWO>
DP_DEVICE void k8_dequant8(const aacg_tables* T, const float* tabq, const unit_view& u, int i, const dpi4& ql, const dpi4& qr,
                           const float* btl, const float* btr, float (&xl)[8], float (&xr)[8])
{
    chan_ctx ccL, ccR;
    ccL.cls = u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE; ccL.max_sfb = u.max_sfb[0]; ccL.gmap = u.gmap[0];
    ccR.cls = u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE; ccR.max_sfb = u.max_sfb[1]; ccR.gmap = u.gmap[1];
    int idxL[2], idxR[2];
    k8_band_idx(tabq, ccL, i, idxL);
    idxR[0] = idxL[0]; idxR[1] = idxL[1];
    if (TWO && (ccR.cls != ccL.cls || ccR.max_sfb != ccL.max_sfb || ccR.gmap != ccL.gmap)) {
        dp_keep_branch();
        k8_band_idx(tabq, ccR, i, idxR);
    }
    dpf2 recL[2], recR[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        recL[k] = *(const dpf2*)(btl + 2 * idxL[k]);
        if (TWO) recR[k] = *(const dpf2*)(btr + 2 * idxR[k]); else { recR[k].x = 0.0f; recR[k].y = 0.0f; }
    }
    dp_lanes g_ms[2], g_is[2];
    bool liveL[2], liveR[2];
    float g_isc[2], sfL[2], sfR[2];
    int big = 0;
    const int iq0 = dp_lds_addr(tabq + AACG_TAB_OFF_IQ_SMALL + 512);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const unsigned fl = __builtin_bit_cast(unsigned, recL[k].y), fr = __builtin_bit_cast(unsigned, recR[k].y);
        g_ms[k] = dp_lanes_where((fl & AACG_BR_FLAG) != 0);
        g_is[k] = dp_lanes_where((fr & AACG_BR_FLAG) != 0);
        const bool is_r = (fr & AACG_BR_FLAG) != 0;
        liveL[k] = (fl & AACG_BR_LIVE) != 0;
        liveR[k] = (fr & AACG_BR_LIVE) != 0;
        sfL[k] = recL[k].x;
        g_isc[k] = recR[k].x;
        sfR[k] = is_r ? 0.0f : recR[k].x;
        dequant4(iq0, sfL[k], __builtin_bit_cast(float, fl & AACG_BR_LIVE), k ? ql.z : ql.x, k ? ql.w : ql.y, *(float (*)[4])&xl[4 * k], big);
        if (TWO) dequant4(iq0, sfR[k], __builtin_bit_cast(float, fr & AACG_BR_LIVE), k ? qr.z : qr.x, k ? qr.w : qr.y, *(float (*)[4])&xr[4 * k], big);
    }
    if (dp_any((big & AACG_OOR_MASK) != 0)) {          /* escape-coded magnitudes: rare */
#pragma unroll
        for (int k = 0; k < 2; k++) {
            dequant4_big(T, liveL[k], sfL[k], k ? ql.z : ql.x, k ? ql.w : ql.y, *(float (*)[4])&xl[4 * k]);
            if (TWO) dequant4_big(T, liveR[k], sfR[k], k ? qr.z : qr.x, k ? qr.w : qr.y, *(float (*)[4])&xr[4 * k]);
        }
    }
    if (TWO) {
        if (dp_lanes_any(g_ms[0] | g_ms[1])) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                dpv2 a0 = v2(xl[4 * k], xl[4 * k + 1]), a1 = v2(xl[4 * k + 2], xl[4 * k + 3]);
                dpv2 b0 = v2(xr[4 * k], xr[4 * k + 1]), b1 = v2(xr[4 * k + 2], xr[4 * k + 3]);
                dp_sumdiff_where(g_ms[k], a0, a1, b0, b1);
                xl[4 * k] = a0[0]; xl[4 * k + 1] = a0[1]; xl[4 * k + 2] = a1[0]; xl[4 * k + 3] = a1[1];
                xr[4 * k] = b0[0]; xr[4 * k + 1] = b0[1]; xr[4 * k + 2] = b1[0]; xr[4 * k + 3] = b1[1];
            }
        }
        if (dp_lanes_any(g_is[0] | g_is[1])) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const dpv2 l0 = v2(xl[4 * k], xl[4 * k + 1]), l1 = v2(xl[4 * k + 2], xl[4 * k + 3]);
                dpv2 r0 = v2(xr[4 * k], xr[4 * k + 1]), r1 = v2(xr[4 * k + 2], xr[4 * k + 3]);
                dp_scale_where(g_is[k], l0, l1, g_isc[k], r0, r1);
                xr[4 * k] = r0[0]; xr[4 * k + 1] = r0[1]; xr[4 * k + 2] = r1[0]; xr[4 * k + 3] = r1[1];
            }
        }
    }
}

/* coefficients 8 lane + 512 i + 0..7 -> the E / O planes of a slot */
DP_DEVICE void k8_stage8(float* slot, int i, const float (&x)[8])
{
    const int k0 = k8_stg(4 * dp_lane() + 256 * i);
    dpf4 ev, od;
    ev.x = x[0]; ev.y = x[2]; ev.z = x[4]; ev.w = x[6];
    od.x = x[1]; od.y = x[3]; od.z = x[5]; od.w = x[7];
    *(dpf4*)(slot + k0) = ev;
    *(dpf4*)(slot + 512 + k0) = od;
}

/* ---- one run per workgroup ---------------------------------------------------------------------------------------------- */
template <int KIND>
DP_DEVICE void imdct_run8_body(const aacg_kparams8& P)
{
    const int TAB_FLOATS = (KIND == AACG_INPUT_QUANT_I16) ? AACG8_TAB_QUANT_FLOATS : AACG8_TAB_F32_FLOATS;
    const int lane = dp_lane(), wave = dp_wave(), tid = dp_tid();
    const aacg_run8* run = P.runs + dp_block();
    float* lds = (float*)dp_lds_fixed<AACG8_LDS_BYTES(TAB_FLOATS)>();
    float* slots = lds;
    float* tabw = lds + AACG_WG_WAVES * AACG8_SLOT_FLOATS;
    const float* tab = tabw;
    const float* tabq = K8_TABQ(tab);
    int* flags = (int*)(tabw + TAB_FLOATS);
    const aacg_win8* W = P.win;
#ifdef AACG_PROFILE
    unsigned long long* trace = P.trace ? P.trace + ((size_t)dp_block() * AACG_WG_WAVES + wave) * 8 : nullptr;
#endif
    K8_TRACE(0);

    /* table loads first (everything behind them may stay in flight while they are copied to LDS): one float4 per thread,
     * the windows' place in aacg_tables skipped */
    const int n4 = TAB_FLOATS / 4, f4 = AACG8_TAB_F32_FLOATS / 4;
    const int tsrc = tid < f4 ? tid : tid + AACG8_WIN_GAP_FLOATS / 4;
    const dpf4 tr = ((const dpf4*)P.tab)[tid < n4 ? tsrc : 0];

    const int n_units = run->n_units;
    const bool two = run->n_ch == 2;
    const int f = two ? wave >> 1 : wave, c = two ? wave & 1 : 0;
    const bool active = f < n_units;
    const int ui = dp_uniform(active ? run->unit[f] : run->unit[0]);
    const unit_view u = load_unit(P.units + ui);
    const int last = n_units - 1;
    /* window fields of the frame before this one (two scalar dwords of its unit record), requested now: behind the transform
     * they were a memory round trip in front of the window loads */
    const uint32_t* pw = (const uint32_t*)(P.units + dp_uniform(active && f > 0 ? run->unit[f - 1] : run->unit[0]));
    const uint32_t pw0 = pw[6], pw1 = pw[10];
    /* earlier frames first; the frame another workgroup waits for (a run's last, when the chain goes on) ahead of them */
    const bool hands_over = active && f == last && run->link_out >= 0;
    dp_setprio(hands_over ? 3 : 3 - (two ? f >> 1 : f >> 2));

    float* slot = slots + wave * AACG8_SLOT_FLOATS;
    float* slot_l = two ? slots + (wave & ~1) * AACG8_SLOT_FLOATS : slot;      /* the pair's left / right slots */
    float* slot_r = slot_l + AACG8_SLOT_FLOATS;

    /* this wave's share of the spectrum: quantised pair -> coefficients 512 c .. of both channels; else its own channel */
    dpi4 ql, qr;
    unsigned mw[2][2];
    dpf4 xa[4];
    auto issue_loads = [&]() {
        if (KIND == AACG_INPUT_QUANT_I16) {
            const int16_t* q0 = (const int16_t*)P.coeffs + (size_t)u.coef_offset * 1024u;
            const aacg_band_meta* m0 = P.meta + u.meta_offset;
            const aacg_band_meta* m1 = m0 + (two ? 1 : 0);
            const int b1 = lane + 64 < AACG_MAX_SECTIONS ? lane + 64 : AACG_MAX_SECTIONS - 1;
            ql = *(const dpi4*)(q0 + 8 * lane + (two ? 512 * c : 0));
            qr = *(const dpi4*)(q0 + 8 * lane + (two ? 1024 + 512 * c : 512));
            mw[0][0] = m0->band[lane]; mw[0][1] = m0->band[b1];
            mw[1][0] = m1->band[lane]; mw[1][1] = m1->band[b1];
        } else {
            const float* xs = (const float*)P.coeffs + (size_t)(u.coef_offset + (uint32_t)c) * 1024u;
#pragma unroll
            for (int i = 0; i < 4; i++) xa[i] = *(const dpf4*)(xs + 4 * lane + 256 * i);
        }
    };
    /* load staggering as in the 16-wave kernels: the run's first frames request their spectra ahead of the table barrier (which
     * is then only released once that data has landed), the others behind it — the first frames see their data after one
     * round trip instead of queueing behind the whole chip's requests */
    const bool early = wave < AACG8_EARLY_WAVES || hands_over;     /* (the frame another workgroup may be waiting for: no later than the first) */
    if (early) issue_loads();
    if (tid < n4) ((dpf4*)tabw)[tid] = tr;
    if (lane == 0) flags[wave] = 0;
    dp_block_sync_lds();                               /* tables and flags are in LDS */
    if (!early) issue_loads();
    K8_TRACE(1);
    if (!active) return;

    /* (selected, not indexed: an array indexed by a run-time value goes to scratch memory) */
    const bool is_short = (c ? u.seq[1] : u.seq[0]) == AACG_EIGHT_SHORT_SEQUENCE;

    /* ---- front end: spectrum -> E / O planes of the channel's slot ---- */
    if (KIND == AACG_INPUT_QUANT_I16) {
        const bool shorts = (u.seq[0] == AACG_EIGHT_SHORT_SEQUENCE) | (two && u.seq[1] == AACG_EIGHT_SHORT_SEQUENCE);
        const bool ms_on = two && (u.flags & AACG_UNIT_COMMON_WINDOW) && (u.flags & AACG_UNIT_MASK_PRESENT);
        const bool mask = (u.flags & AACG_UNIT_MASK_PRESENT) != 0;
        if (two) {
            /* the band records live where only this wave writes later: its halves of the right channel's planes */
            float* btl = slot_r + 256 * c;
            float* btr = slot_r + 512 + 256 * c;
            if (shorts) { dp_keep_branch(); k8_prepare_bands<0, 2, true>(mw, tabq, ms_on, mask, btl, btr); }
            else {
                k8_prepare_bands<0, 1, true>(mw, tabq, ms_on, mask, btl, btr);
                dpf2 none; none.x = none.y = 0.0f;
                *(dpf2*)(btl + 2 * (64 + lane)) = none;
                *(dpf2*)(btr + 2 * (64 + lane)) = none;
            }
            dp_wave_sync();
            float xl[8], xr[8];
            k8_dequant8<true>(P.tab, tabq, u, c, ql, qr, btl, btr, xl, xr);
            dp_wave_sync();                            /* band records dead */
            k8_stage8(slot_l, c, xl);
            k8_stage8(slot_r, c, xr);
            dp_wave_sync();
            if (lane == 0) dp_flag_set(&flags[wave], K8_PHASE_STAGED);
            dp_flag_wait_ge(&flags[wave ^ 1], K8_PHASE_STAGED);    /* the other half of this channel's spectrum */
        } else {
            float* btl = slot + 256;                   /* untouched by the first pass's staging */
            if (shorts) { dp_keep_branch(); k8_prepare_bands<0, 2, false>(mw, tabq, false, mask, btl, btl); }
            else {
                k8_prepare_bands<0, 1, false>(mw, tabq, false, mask, btl, btl);
                dpf2 none; none.x = none.y = 0.0f;
                *(dpf2*)(btl + 2 * (64 + lane)) = none;
            }
            dp_wave_sync();
            float x0[8], x1[8], dummy[8];
            k8_dequant8<false>(P.tab, tabq, u, 0, ql, ql, btl, btl, x0, dummy);
            k8_dequant8<false>(P.tab, tabq, u, 1, qr, qr, btl, btl, x1, dummy);
            dp_wave_sync();                            /* band records dead */
            k8_stage8(slot, 0, x0);
            k8_stage8(slot, 1, x1);
            dp_wave_sync();
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {                  /* coefficients 4 lane + 256 i + 0..3 */
            const int k0 = k8_stg(2 * lane + 128 * i);
            dpf2 ev, od;
            ev.x = xa[i].x; ev.y = xa[i].z; od.x = xa[i].y; od.y = xa[i].w;
            *(dpf2*)(slot + k0) = ev;
            *(dpf2*)(slot + 512 + k0) = od;
        }
        dp_wave_sync();
    }

    K8_TRACE(2);
    /* ---- transform: the rotated FFT output of this channel stays in its slot ---- */
    if (is_short) { dp_keep_branch(); k8_short(tab, slot); }
    else          k8_long(tab, slot);
    dp_wave_sync();
    if (lane == 0) dp_flag_set(&flags[wave], K8_PHASE_DUMPED);
    K8_TRACE(3);

    /* ---- finishing: wave c of a pair takes samples 512 c .. 512 c + 511 of both channels; a single channel all 1024 ---- */
    if (two) dp_flag_wait_ge(&flags[wave ^ 1], K8_PHASE_DUMPED);
    const int C = u.n_out_ch;
    const int n0 = two ? 512 * c : 0;
    k8_src cur;
    cur.lds[0] = slot_l; cur.lds[1] = slot_r; cur.glob[0] = cur.glob[1] = nullptr;
#pragma unroll
    for (int k = 0; k < 2; k++) { cur.ch[k].seq = u.seq[k]; cur.ch[k].shape = u.shape[k]; cur.ch[k].shape_prev = u.shape_prev[k]; }
    const unsigned long long tag = P.epoch << 2;

    /* a run's last frame, when the chain goes on in another workgroup: publish the windowed tail, or — if that workgroup was
     * here first and left its windowed first half — finish its frame */
    if (f == last && run->link_out >= 0) {
        dp_keep_branch();
        unsigned long long* st = P.rv_state + (size_t)run->link_out * AACG8_RV_STATE_WORDS + c;
        float* data = P.rv_data + (size_t)run->link_out * AACG8_RV_DATA_FLOATS;
        unsigned long long seen = dp_first_u64(dp_g_load_u64(st));
        bool theirs = seen == (tag | AACG8_RV_HEAD);
        if (!theirs) {
            k8_export_any<true, true>(W, cur, two ? 2 : 1, !two, c, data, data + 1024);
            dp_vm_drain();
            dp_wave_sync();
            bool won = true;
            if (lane == 0) won = dp_g_cas_u64(st, seen, tag | AACG8_RV_TAIL);
            theirs = dp_first_u64(won ? 0ull : 1ull) != 0ull;        /* lost: the other side published meanwhile */
        }
        if (theirs) {
            const unit_view su = load_unit(P.units + dp_uniform(run->succ_unit));
            k8_src hs;
            hs.lds[0] = hs.lds[1] = nullptr; hs.glob[0] = data + 2048; hs.glob[1] = data + 2048 + 1024;
            hs.ch[0] = cur.ch[0]; hs.ch[1] = cur.ch[1];
            float* spcm = P.pcm + su.pcm_offset + su.channel;
            const bool sfast = two && su.n_out_ch == 2 && ((su.pcm_offset | (uint32_t)su.channel) & 3u) == 0 &&
                               cur.ch[0].seq != AACG_EIGHT_SHORT_SEQUENCE && cur.ch[0].seq == cur.ch[1].seq && cur.ch[0].shape == cur.ch[1].shape;
            if (sfast) {
                k8_wins wins;
                k8_stereo_windows(W, cur, hs, c, wins);          /* (only the tail windows are used) */
                k8_finish_stereo_long(wins, cur, hs, spcm, c);
            } else if (two) { dp_keep_branch(); k8_finish_general<2, 8>(W, cur, hs, spcm, su.n_out_ch, n0); }
            else            { dp_keep_branch(); k8_finish_general<1, 16>(W, cur, hs, spcm, su.n_out_ch, n0); }
        }
    }
    /* the chain's last frame in this launch: its windowed tail is the new overlap state (planar in HBM) */
    if (f == last && run->link_out < 0) {
        dp_keep_branch();
        float* o0 = P.overlap + (P.flip ? run->ov_a[0] : run->ov_b[0]);
        float* o1 = P.overlap + (P.flip ? run->ov_a[1] : run->ov_b[1]);
        k8_export_any<true, false>(W, cur, two ? 2 : 1, !two, c, o0, o1);
    }

    /* this frame's own output: the tail of the frame before it + its windowed first half */
    float* pcm = P.pcm + u.pcm_offset + u.channel;
    k8_src prev;
    prev.glob[0] = prev.glob[1] = nullptr;
    prev.lds[0] = slot_l - (two ? 2 : 1) * AACG8_SLOT_FLOATS; prev.lds[1] = slot_r - 2 * AACG8_SLOT_FLOATS;
    prev.ch[0] = cur.ch[0]; prev.ch[1] = cur.ch[1];
    const bool from_state = f == 0 && run->link_in < 0, from_link = f == 0 && run->link_in >= 0;
    if (f > 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) { const uint32_t ci = k ? pw1 : pw0; prev.ch[k].seq = (int)(ci & 0xffu); prev.ch[k].shape = (int)((ci >> 8) & 0xffu); prev.ch[k].shape_prev = (int)((ci >> 16) & 0xffu); }
    }
    /* which pass finishes the frame is known from window fields alone: decide, and request the fast pass's windows, before waiting */
    const bool long_l = cur.ch[0].seq != AACG_EIGHT_SHORT_SEQUENCE && (f == 0 || prev.ch[0].seq != AACG_EIGHT_SHORT_SEQUENCE);
    const bool long_r = cur.ch[1].seq != AACG_EIGHT_SHORT_SEQUENCE && (f == 0 || prev.ch[1].seq != AACG_EIGHT_SHORT_SEQUENCE);
    /* the same window tables for both channels (a common window, or equal by value) */
    const bool same_tables = cur.ch[0].seq == cur.ch[1].seq && cur.ch[0].shape_prev == cur.ch[1].shape_prev &&
                             (f == 0 || (prev.ch[0].seq == prev.ch[1].seq && prev.ch[0].shape == prev.ch[1].shape));
    const bool fast = two && C == 2 && long_l && long_r && same_tables && ((u.pcm_offset | (uint32_t)u.channel) & 3u) == 0;
    if (fast && f > 0) {
        /* the common case, kept apart from the rest so that its window registers live nowhere else */
        k8_wins wins;
        k8_stereo_windows(W, prev, cur, c, wins);
        dp_flag_wait_ge(&flags[wave - 2], K8_PHASE_DUMPED);
        dp_flag_wait_ge(&flags[(wave ^ 1) - 2], K8_PHASE_DUMPED);
        K8_TRACE(4);
        k8_finish_stereo_long(wins, prev, cur, pcm, c);
        K8_TRACE(5);
        return;
    }
    if (f > 0) {
        dp_flag_wait_ge(&flags[wave - (two ? 2 : 1)], K8_PHASE_DUMPED);
        if (two) dp_flag_wait_ge(&flags[(wave ^ 1) - 2], K8_PHASE_DUMPED);
    } else if (from_state) {
        /* first frame of its chain in this launch: the overlap state (filter_bank.js:38-41) */
        prev.lds[0] = prev.lds[1] = nullptr;
        prev.glob[0] = P.overlap + (P.flip ? run->ov_b[0] : run->ov_a[0]);
        prev.glob[1] = P.overlap + (P.flip ? run->ov_b[1] : run->ov_a[1]);
    } else if (from_link) {
        /* first frame of a later run: the tail the run before it published — or, if that is not there yet, leave the windowed
         * first half for it and go */
        dp_keep_branch();
        unsigned long long* st = P.rv_state + (size_t)run->link_in * AACG8_RV_STATE_WORDS + c;
        float* data = P.rv_data + (size_t)run->link_in * AACG8_RV_DATA_FLOATS;
        unsigned long long seen = dp_first_u64(dp_g_load_u64(st));
        bool theirs = seen == (tag | AACG8_RV_TAIL);
        if (!theirs) {
            k8_export_any<false, true>(W, cur, two ? 2 : 1, !two, c, data + 2048, data + 2048 + 1024);
            dp_vm_drain();
            dp_wave_sync();
            bool won = true;
            if (lane == 0) won = dp_g_cas_u64(st, seen, tag | AACG8_RV_HEAD);
            theirs = dp_first_u64(won ? 0ull : 1ull) != 0ull;
            if (!theirs) return;                       /* the run before this one finishes the frame */
        }
        prev.lds[0] = prev.lds[1] = nullptr;
        prev.glob[0] = data; prev.glob[1] = data + 1024;
    }
    K8_TRACE(4);
    if (fast) {
        dp_keep_branch();
        k8_wins wins;
        k8_stereo_windows(W, prev, cur, c, wins);
        k8_finish_stereo_long(wins, prev, cur, pcm, c);
    } else if (two) {
        dp_keep_branch();
        k8_finish_general<2, 8>(W, prev, cur, pcm, C, n0);
    } else {
        dp_keep_branch();
        k8_finish_general<1, 16>(W, prev, cur, pcm, C, n0);
    }
    K8_TRACE(5);
}

#endif /* AACG_KERNELS8_H */
