/*
 * aac_oracle.c — CPU restatement of aac.js's per-frame transform path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see aac_oracle.h).  Parity status: PINNED against outputs of
 * the reference itself (tests/golden, tests/test_oracle_golden.py): tables, FFT, IMDCT and
 * the whole filterbank / dequant / MS / IS chain are reproduced bit-for-bit.
 *
 * Every function cites the reference file:line it follows.  Rounding model: all
 * arithmetic in double, rounded to float exactly where the reference stores into a
 * Float32Array.  Build: gcc -O2 -ffp-contract=off (an FMA-contracted build is a
 * different function).
 */
#include "aac_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------ */
/* tables                                                                                 */
/* ------------------------------------------------------------------------------------ */

static int    g_ready;
static float  g_iq[8191];             /* tables.js:182-191 */
static float  g_sf[428];              /* tables.js:168-176 */
static float  g_sine_long[1024], g_sine_short[128];   /* filter_bank.js:46-52,81-82 */
static float  g_kbd_long[1024],  g_kbd_short[128];    /* filter_bank.js:54-79,83-84 */
static float  g_roots_long[512][3];   /* fft.js:82-103 */
static float  g_roots_short[64][2];   /* fft.js:59-80  */
static double g_mdct_long[512][2];    /* mdct_tables.js:21-534  (N = 2048) */
static double g_mdct_short[64][2];    /* mdct_tables.js:536-601 (N = 256)  */

/* tables.js:34-155 as (width, repeat) runs; -1 terminates.  Index = sampleIndex 0..11. */
static const int8_t SWB_LONG_RLE[12][32] = {
    {4,14, 8,5, 12,5, 16,2, 24,1, 28,1, 36,1, 44,1, 64,11, -1},
    {4,14, 8,5, 12,5, 16,2, 24,1, 28,1, 36,1, 44,1, 64,11, -1},
    {4,14, 8,4, 12,3, 16,3, 20,1, 24,2, 28,1, 36,1, 40,18, -1},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,19, 96,1, -1},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,19, 96,1, -1},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,22, -1},
    {4,11, 8,10, 12,4, 16,3, 20,2, 24,2, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,2, 64,5, -1},
    {4,11, 8,10, 12,4, 16,3, 20,2, 24,2, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,2, 64,5, -1},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, -1},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, -1},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, -1},
    {12,13, 16,7, 20,4, 24,3, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,1, 56,1, 60,1, 64,1, 80,1, -1},
};
static const int8_t SWB_SHORT_RLE[12][12] = {
    {4,6, 8,3, 16,1, 28,1, 36,1, -1},
    {4,6, 8,3, 16,1, 28,1, 36,1, -1},
    {4,6, 8,3, 16,1, 28,1, 36,1, -1},
    {4,5, 8,3, 12,3, 16,3, -1},
    {4,5, 8,3, 12,3, 16,3, -1},
    {4,5, 8,3, 12,3, 16,3, -1},
    {4,7, 8,3, 12,2, 16,2, 20,1, -1},
    {4,7, 8,3, 12,2, 16,2, 20,1, -1},
    {4,8, 8,2, 12,2, 16,1, 20,2, -1},
    {4,8, 8,2, 12,2, 16,1, 20,2, -1},
    {4,8, 8,2, 12,2, 16,1, 20,2, -1},
    {4,7, 8,4, 12,1, 16,1, 20,2, -1},
};

static uint16_t g_swb_long[12][64];
static uint16_t g_swb_short[12][16];
static int      g_swb_long_count[12], g_swb_short_count[12];

static int expand_rle(const int8_t* rle, uint16_t* off)
{
    int n = 0, pos = 0;
    off[0] = 0;
    for (int i = 0; rle[i] >= 0; i += 2)
        for (int r = 0; r < rle[i + 1]; r++) { pos += rle[i]; off[++n] = (uint16_t)pos; }
    return n;
}

/* filter_bank.js:46-52 */
static void gen_sine(float* d, int len)
{
    for (int i = 0; i < len; i++)
        d[i] = (float)sin((i + 0.5) * (M_PI / (2.0 * len)));
}

/* filter_bank.js:54-79: note the running sum is read back from a Float32Array (f[n]),
 * while `sum` itself stays double and gets +1 after the loop. */
static void gen_kbd(float* out, double alpha, int len)
{
    double PIN = M_PI / len, sum = 0.0;
    double alpha2 = (alpha * PIN) * (alpha * PIN);
    float* f = (float*)malloc(sizeof(float) * (size_t)len);
    for (int n = 0; n < len; n++) {
        double tmp = (double)n * (double)(len - n) * alpha2, bessel = 1.0;
        for (int j = 50; j > 0; j--)
            bessel = bessel * tmp / (double)(j * j) + 1.0;
        sum += bessel;
        f[n] = (float)sum;
    }
    sum += 1.0;
    for (int n = 0; n < len; n++)
        out[n] = (float)sqrt((double)f[n] / sum);
    free(f);
}

/* fft.js:82-103: every component goes through a Float32Array and is fed back. */
static void gen_roots_long(float (*f)[3], int len)
{
    double t = 2.0 * M_PI / len, cosT = cos(t), sinT = sin(t);
    f[0][0] = 1.0f; f[0][1] = 0.0f; f[0][2] = 0.0f;
    for (int i = 1; i < len; i++) {
        f[i][0] = (float)((double)f[i - 1][0] * cosT + (double)f[i - 1][2] * sinT);
        f[i][2] = (float)((double)f[i - 1][2] * cosT - (double)f[i - 1][0] * sinT);
        f[i][1] = -f[i][2];
    }
}

/* fft.js:59-80: lastImag is a plain JS double. */
static void gen_roots_short(float (*f)[2], int len)
{
    double t = 2.0 * M_PI / len, cosT = cos(t), sinT = sin(t), lastImag = 0.0;
    f[0][0] = 1.0f; f[0][1] = 0.0f;
    for (int i = 1; i < len; i++) {
        f[i][0] = (float)((double)f[i - 1][0] * cosT + lastImag * sinT);
        lastImag = lastImag * cosT - (double)f[i - 1][0] * sinT;
        f[i][1] = (float)(-lastImag);
    }
}

/* mdct_tables.js holds sqrt(2/N) * (cos, sin)(2*pi*(k + 1/8)/N) written with 15 decimals;
 * the literal is what the reference computes with, so go through the same text form. */
static double round15(double v)
{
    char txt[64];
    snprintf(txt, sizeof txt, "%.15f", v);
    return strtod(txt, NULL);
}

static void gen_mdct(double (*t)[2], int N)
{
    double scale = sqrt(2.0 / N);
    for (int k = 0; k < N / 4; k++) {
        double a = 2.0 * M_PI * (k + 0.125) / N;
        t[k][0] = round15(scale * cos(a));
        t[k][1] = round15(scale * sin(a));
    }
}

/* the reference's FFT roots come from a float32 recurrence (fft.js:59-103) that drifts from the exact roots
 * of unity (<= 8.8e-7 for 512 points).  The oracle follows the recurrence — that is what pins it bit for bit to aac.js;
 * orc_set_fft_roots(1) swaps in correctly rounded roots (what the GPU engine uses) so that tests can measure how much of
 * the engine's distance from the reference is that drift and how much is its own arithmetic.  0 restores the recurrence. */
void orc_set_fft_roots(int exact)
{
    orc_init();
    if (!exact) { gen_roots_long(g_roots_long, 512); gen_roots_short(g_roots_short, 64); return; }
    for (int i = 0; i < 512; i++) {
        const double a = 2.0 * M_PI * i / 512.0;
        g_roots_long[i][0] = (float)cos(a); g_roots_long[i][1] = (float)sin(a); g_roots_long[i][2] = -g_roots_long[i][1];
    }
    for (int i = 0; i < 64; i++) {
        const double a = 2.0 * M_PI * i / 64.0;
        g_roots_short[i][0] = (float)cos(a); g_roots_short[i][1] = (float)sin(a);
    }
}

void orc_init(void)
{
    if (g_ready) return;
    for (int i = 0; i < 8191; i++) g_iq[i] = (float)pow((double)i, 4.0 / 3.0);   /* tables.js:186-188 */
    for (int i = 0; i < 428; i++)  g_sf[i] = (float)pow(2.0, (i - 200) / 4.0);   /* tables.js:171-173 */
    gen_sine(g_sine_long, 1024);
    gen_sine(g_sine_short, 128);
    gen_kbd(g_kbd_long, 4.0, 1024);      /* filter_bank.js:83 */
    gen_kbd(g_kbd_short, 6.0, 128);      /* filter_bank.js:84 */
    gen_roots_long(g_roots_long, 512);
    gen_roots_short(g_roots_short, 64);
    gen_mdct(g_mdct_long, 2048);
    gen_mdct(g_mdct_short, 256);
    for (int s = 0; s < 12; s++) {
        g_swb_long_count[s]  = expand_rle(SWB_LONG_RLE[s],  g_swb_long[s]);
        g_swb_short_count[s] = expand_rle(SWB_SHORT_RLE[s], g_swb_short[s]);
    }
    g_ready = 1;
}

size_t orc_get_table_f32(int which, float* dst, size_t n)
{
    const float* src; size_t cnt;
    orc_init();
    switch (which) {
    case 0: src = g_iq;                 cnt = 8191;    break;
    case 1: src = g_sf;                 cnt = 428;     break;
    case 2: src = g_sine_long;          cnt = 1024;    break;
    case 3: src = g_kbd_long;           cnt = 1024;    break;
    case 4: src = g_sine_short;         cnt = 128;     break;
    case 5: src = g_kbd_short;          cnt = 128;     break;
    case 6: src = &g_roots_long[0][0];  cnt = 512 * 3; break;
    case 7: src = &g_roots_short[0][0]; cnt = 64 * 2;  break;
    default: return 0;
    }
    if (dst) memcpy(dst, src, sizeof(float) * (n < cnt ? n : cnt));
    return cnt;
}

size_t orc_get_table_f64(int which, double* dst, size_t n)
{
    const double* src; size_t cnt;
    orc_init();
    switch (which) {
    case 0: src = &g_mdct_long[0][0];  cnt = 512 * 2; break;
    case 1: src = &g_mdct_short[0][0]; cnt = 64 * 2;  break;
    default: return 0;
    }
    if (dst) memcpy(dst, src, sizeof(double) * (n < cnt ? n : cnt));
    return cnt;
}

int orc_get_swb_offsets(int sample_index, int is_long, uint16_t* dst)
{
    orc_init();
    if (sample_index < 0 || sample_index > 11) return 0;
    int cnt = is_long ? g_swb_long_count[sample_index] : g_swb_short_count[sample_index];
    memcpy(dst, is_long ? g_swb_long[sample_index] : g_swb_short[sample_index],
           sizeof(uint16_t) * (size_t)(cnt + 1));
    return cnt;
}

/* ------------------------------------------------------------------------------------ */
/* FFT, fft.js:105-192 (forward = false: imOffset 1, scale 1)                            */
/* ------------------------------------------------------------------------------------ */

void orc_fft_inverse(int length, float* buf)
{
    float (*in)[2] = (float (*)[2])buf;
    float rev[512][2];
    orc_init();

    /* bit-reversal, fft.js:113-125 */
    int ii = 0;
    for (int i = 0; i < length; i++) {
        rev[i][0] = in[ii][0];
        rev[i][1] = in[ii][1];
        int k = length >> 1;
        while (ii >= k && k > 0) { ii -= k; k >>= 1; }
        ii += k;
    }
    for (int i = 0; i < length; i++) { in[i][0] = rev[i][0]; in[i][1] = rev[i][1]; }

    /* bottom base-4 round, fft.js:140-170: every temporary lives in a Float32Array */
    for (int i = 0; i < length; i += 4) {
        float a0 = in[i][0] + in[i + 1][0],     a1 = in[i][1] + in[i + 1][1];
        float b0 = in[i + 2][0] + in[i + 3][0], b1 = in[i + 2][1] + in[i + 3][1];
        float c0 = in[i][0] - in[i + 1][0],     c1 = in[i][1] - in[i + 1][1];
        float d0 = in[i + 2][0] - in[i + 3][0], d1 = in[i + 2][1] - in[i + 3][1];
        in[i][0] = a0 + b0;     in[i][1] = a1 + b1;
        in[i + 2][0] = a0 - b0; in[i + 2][1] = a1 - b1;
        float e10 = c0 - d1, e11 = c1 + d0;
        float e20 = c0 + d1, e21 = c1 - d0;
        in[i + 1][0] = e10; in[i + 1][1] = e11;      /* !forward branch, fft.js:164-168 */
        in[i + 3][0] = e20; in[i + 3][1] = e21;
    }

    /* iterations from bottom to top, fft.js:173-191: zRe/zIm are JS doubles */
    for (int i = 4; i < length; i <<= 1) {
        int shift = i << 1, m = length / shift;
        for (int j = 0; j < length; j += shift) {
            for (int k = 0; k < i; k++) {
                int km = k * m;
                double rootRe, rootIm;
                if (length == 512) { rootRe = g_roots_long[km][0];  rootIm = g_roots_long[km][1]; }
                else               { rootRe = g_roots_short[km][0]; rootIm = g_roots_short[km][1]; }
                double xr = in[i + j + k][0], xi = in[i + j + k][1];
                double zRe = xr * rootRe - xi * rootIm;
                double zIm = xr * rootIm + xi * rootRe;
                double lr = in[j + k][0], li = in[j + k][1];
                in[i + j + k][0] = (float)(lr - zRe);
                in[i + j + k][1] = (float)(li - zIm);
                in[j + k][0] = (float)(lr + zRe);
                in[j + k][1] = (float)(li + zIm);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* IMDCT, mdct.js:62-115                                                                  */
/* ------------------------------------------------------------------------------------ */

void orc_imdct(int N, const float* input, float* output)
{
    int N2 = N >> 1, N4 = N >> 2, N8 = N >> 3;
    float buf[512][2];
    orc_init();
    double (*sincos)[2] = (N == 2048) ? g_mdct_long : g_mdct_short;

    /* pre-IFFT complex multiplication, mdct.js:73-76 */
    for (int k = 0; k < N4; k++) {
        double x0 = input[2 * k], x1 = input[N2 - 1 - 2 * k];
        buf[k][1] = (float)((x0 * sincos[k][0]) + (x1 * sincos[k][1]));
        buf[k][0] = (float)((x1 * sincos[k][0]) - (x0 * sincos[k][1]));
    }

    orc_fft_inverse(N4, &buf[0][0]);               /* mdct.js:79 */

    /* post-IFFT complex multiplication, mdct.js:82-87 */
    for (int k = 0; k < N4; k++) {
        double t0 = buf[k][0], t1 = buf[k][1];
        buf[k][1] = (float)((t1 * sincos[k][0]) + (t0 * sincos[k][1]));
        buf[k][0] = (float)((t0 * sincos[k][0]) - (t1 * sincos[k][1]));
    }

    /* reordering, mdct.js:90-114 */
    for (int k = 0; k < N8; k += 2) {
        output[2 * k]     = buf[N8 + k][1];
        output[2 + 2 * k] = buf[N8 + 1 + k][1];
        output[1 + 2 * k] = -buf[N8 - 1 - k][0];
        output[3 + 2 * k] = -buf[N8 - 2 - k][0];

        output[N4 + 2 * k]     = buf[k][0];
        output[N4 + 2 + 2 * k] = buf[1 + k][0];
        output[N4 + 1 + 2 * k] = -buf[N4 - 1 - k][1];
        output[N4 + 3 + 2 * k] = -buf[N4 - 2 - k][1];

        output[N2 + 2 * k]     = buf[N8 + k][0];
        output[N2 + 2 + 2 * k] = buf[N8 + 1 + k][0];
        output[N2 + 1 + 2 * k] = -buf[N8 - 1 - k][1];
        output[N2 + 3 + 2 * k] = -buf[N8 - 2 - k][1];

        output[N2 + N4 + 2 * k]     = -buf[k][1];
        output[N2 + N4 + 2 + 2 * k] = -buf[1 + k][1];
        output[N2 + N4 + 1 + 2 * k] = buf[N4 - 1 - k][0];
        output[N2 + N4 + 3 + 2 * k] = buf[N4 - 2 - k][0];
    }
}

/* ------------------------------------------------------------------------------------ */
/* filterbank, filter_bank.js:88-204                                                      */
/* ------------------------------------------------------------------------------------ */

/* f32 store of  a + b*w  evaluated in double, filter_bank.js:110 etc. */
static inline float madd(float a, float b, float w)
{
    return (float)((double)a + ((double)b * (double)w));
}

void orc_filterbank(int window_sequence, int window_shape, int window_shape_prev,
                    const float* input, float* output, float* overlap)
{
    enum { length = 1024, shortLen = 128, mid = 448, trans = 64 };
    float buf[2048];
    orc_init();
    const float* longWindows      = window_shape      ? g_kbd_long  : g_sine_long;
    const float* shortWindows     = window_shape      ? g_kbd_short : g_sine_short;
    const float* longWindowsPrev  = window_shape_prev ? g_kbd_long  : g_sine_long;
    const float* shortWindowsPrev = window_shape_prev ? g_kbd_short : g_sine_short;

    switch (window_sequence) {
    case AACG_ONLY_LONG_SEQUENCE:                       /* filter_bank.js:105-118 */
        orc_imdct(2048, input, buf);
        for (int i = 0; i < length; i++)
            output[i] = madd(overlap[i], buf[i], longWindowsPrev[i]);
        for (int i = 0; i < length; i++)
            overlap[i] = buf[length + i] * longWindows[length - 1 - i];
        break;

    case AACG_LONG_START_SEQUENCE:                      /* filter_bank.js:120-141 */
        orc_imdct(2048, input, buf);
        for (int i = 0; i < length; i++)
            output[i] = madd(overlap[i], buf[i], longWindowsPrev[i]);
        for (int i = 0; i < mid; i++)
            overlap[i] = buf[length + i];
        for (int i = 0; i < shortLen; i++)
            overlap[mid + i] = buf[length + mid + i] * shortWindows[shortLen - i - 1];
        for (int i = 0; i < mid; i++)
            overlap[mid + shortLen + i] = 0.0f;
        break;

    case AACG_EIGHT_SHORT_SEQUENCE:                     /* filter_bank.js:143-178 */
        for (int i = 0; i < 8; i++)
            orc_imdct(256, input + i * shortLen, buf + 2 * i * shortLen);
        for (int i = 0; i < mid; i++)
            output[i] = overlap[i];
        for (int i = 0; i < shortLen; i++) {
            output[mid + i] = madd(overlap[mid + i], buf[i], shortWindowsPrev[i]);
            for (int j = 1; j <= 4; j++) {
                if (j == 4 && i >= trans) break;
                /* ov + (b1*w1) + (b2*w2), left to right in double, filter_bank.js:155-160 */
                double v = (double)overlap[mid + shortLen * j + i]
                         + ((double)buf[shortLen * (2 * j - 1) + i] * (double)shortWindows[shortLen - 1 - i]);
                v = v + ((double)buf[shortLen * 2 * j + i] * (double)shortWindows[i]);
                output[mid + j * shortLen + i] = (float)v;
            }
        }
        for (int i = 0; i < shortLen; i++) {
            for (int j = 4; j <= 7; j++) {
                if (j == 4 && i < trans) continue;
                double v = ((double)buf[shortLen * (2 * j - 1) + i] * (double)shortWindows[shortLen - 1 - i])
                         + ((double)buf[shortLen * 2 * j + i] * (double)shortWindows[i]);
                overlap[mid + j * shortLen + i - length] = (float)v;
            }
            overlap[mid + 8 * shortLen + i - length] = buf[shortLen * 15 + i] * shortWindows[shortLen - 1 - i];
        }
        for (int i = 0; i < mid; i++)
            overlap[mid + shortLen + i] = 0.0f;
        break;

    case AACG_LONG_STOP_SEQUENCE:                       /* filter_bank.js:180-202 */
        orc_imdct(2048, input, buf);
        for (int i = 0; i < mid; i++)
            output[i] = overlap[i];
        for (int i = 0; i < shortLen; i++)
            output[mid + i] = madd(overlap[mid + i], buf[mid + i], shortWindowsPrev[i]);
        for (int i = 0; i < mid; i++)
            output[mid + shortLen + i] = overlap[mid + shortLen + i] + buf[mid + shortLen + i];
        for (int i = 0; i < length; i++)
            overlap[i] = buf[length + i] * longWindows[length - 1 - i];
        break;
    }
}

/* ------------------------------------------------------------------------------------ */
/* spectral reconstruction                                                                 */
/* ------------------------------------------------------------------------------------ */

static inline int meta_bt(uint16_t m)   { return m >> AACG_META_BT_SHIFT; }
static inline float meta_sf(uint16_t m)
{
    float v = g_sf[m & AACG_META_SF_MASK];
    return (m & AACG_META_NEGATE) ? -v : v;
}

static const uint16_t* swb_of(int sample_index, const aacg_chan_info* info)
{
    /* ics.js:301,307: swbOffsets chosen by window sequence */
    return info->window_sequence == AACG_EIGHT_SHORT_SEQUENCE ? g_swb_short[sample_index]
                                                              : g_swb_long[sample_index];
}

/* ics.js:203-261 with Huffman.decodeSpectralData replaced by the q[] array.  The
 * reference starts from a fresh zeroed Float32Array (ics.js:29), so coefficients at and
 * above swbOffsets[maxSFB] stay +0. */
int orc_dequant(int sample_index, const aacg_chan_info* info, const aacg_band_meta* meta,
                const int16_t* q, float* data)
{
    return orc_dequant_pns(sample_index, info, meta, q, AACG_PNS_REFERENCE, data);
}

/* pns_mode == AACG_PNS_SPEC: NOISE_BT bands as ics.js:228-243 was meant to fill them — the generator
 * `randomState = randomState * 1664525 + 1013904223` (ics.js:234 misplaces a parenthesis and degenerates),
 * restarted at 0x1F2E3D4C for every channel of every frame because the reference builds a fresh ICStream per
 * frame (decoder.js:145,153), values normalised to the band's energy scalefactor per window.  NOT pinned by the
 * reference (it never produces this output); tests check it against an independent numpy restatement. */
int orc_dequant_pns(int sample_index, const aacg_chan_info* info, const aacg_band_meta* meta,
                    const int16_t* q, int pns_mode, float* data)
{
    orc_init();
    int32_t randomState = 0x1F2E3D4C;                                            /* ics.js:31 */
    const uint16_t* offsets = swb_of(sample_index, info);
    memset(data, 0, sizeof(float) * 1024);
    int groupOff = 0, idx = 0;
    for (int g = 0; g < info->group_count; g++) {
        int groupLen = info->group_len[g];
        for (int sfb = 0; sfb < info->max_sfb; sfb++, idx++) {
            int hcb = meta_bt(meta->band[idx]);
            int off = groupOff + offsets[sfb];
            int width = offsets[sfb + 1] - offsets[sfb];
            if (hcb == AACG_ZERO_BT || hcb == AACG_INTENSITY_BT || hcb == AACG_INTENSITY_BT2) {
                for (int group = 0; group < groupLen; group++, off += 128)      /* ics.js:222-227 */
                    for (int i = off; i < off + width; i++) data[i] = 0.0f;
            } else if (hcb == AACG_NOISE_BT) {
                if (pns_mode != AACG_PNS_SPEC)
                    return AACG_ERR_UNSUPPORTED; /* ics.js:228-243 degenerates to NaN, SURVEY §8a row 4 */
                const float sf = meta_sf(meta->band[idx]);
                for (int group = 0; group < groupLen; group++, off += 128) {    /* ics.js:230-242, repaired */
                    double energy = 0.0;
                    for (int k = 0; k < width; k++) {
                        randomState = (int32_t)((uint32_t)randomState * 1664525u + 1013904223u);
                        data[off + k] = (float)randomState;
                        energy += (double)data[off + k] * (double)data[off + k];
                    }
                    const double scale = (double)sf / sqrt(energy);
                    for (int k = 0; k < width; k++) data[off + k] = (float)((double)data[off + k] * scale);
                }
            } else {
                float sf = meta_sf(meta->band[idx]);
                for (int group = 0; group < groupLen; group++, off += 128) {    /* ics.js:244-256 */
                    for (int k = 0; k < width; k++) {
                        int v = q[off + k];
                        float x;
                        if (v > 0) x = (v < 8191) ? g_iq[v] : NAN;              /* IQ_TABLE[8191+] is undefined */
                        else       x = (-v < 8191) ? -g_iq[-v] : NAN;           /* q == 0 gives -0, as in JS    */
                        x = x * sf;
                        data[off + k] = x;
                    }
                }
            }
        }
        groupOff += groupLen << 7;
    }
    return AACG_OK;
}

/* decoder.js:379-404 */
void orc_process_ms(int sample_index, const aacg_unit_desc* u,
                    const aacg_band_meta* meta_l, const aacg_band_meta* meta_r,
                    float* left, float* right)
{
    orc_init();
    const aacg_chan_info* info = &u->ch[0];            /* element.left.info */
    const uint16_t* offsets = swb_of(sample_index, info);
    int groupOff = 0, idx = 0;
    for (int g = 0; g < info->group_count; g++) {
        for (int i = 0; i < info->max_sfb; i++, idx++) {
            int used = (meta_l->band[idx] & AACG_META_MS_USED) != 0;
            if (used && meta_bt(meta_l->band[idx]) < AACG_NOISE_BT && meta_bt(meta_r->band[idx]) < AACG_NOISE_BT) {
                for (int w = 0; w < info->group_len[g]; w++) {
                    int off = groupOff + w * 128 + offsets[i];
                    for (int j = 0; j < offsets[i + 1] - offsets[i]; j++) {
                        float t = left[off + j] - right[off + j];
                        left[off + j] = left[off + j] + right[off + j];
                        right[off + j] = t;
                    }
                }
            }
        }
        groupOff += info->group_len[g] * 128;
    }
}

/* decoder.js:337-376.  sectEnd only drives the iteration there; sections are runs of one
 * band type (ics.js:83-116) so a per-band test is the same function. */
void orc_process_is(int sample_index, const aacg_unit_desc* u,
                    const aacg_band_meta* meta_l, const aacg_band_meta* meta_r,
                    const float* left, float* right)
{
    orc_init();
    const aacg_chan_info* info = &u->ch[1];            /* element.right.info */
    const uint16_t* offsets = swb_of(sample_index, info);
    int groupOff = 0, idx = 0;
    for (int g = 0; g < info->group_count; g++) {
        for (int i = 0; i < info->max_sfb; i++, idx++) {
            int bt = meta_bt(meta_r->band[idx]);
            if (bt == AACG_INTENSITY_BT || bt == AACG_INTENSITY_BT2) {
                int c = (bt == AACG_INTENSITY_BT) ? 1 : -1;
                if (u->flags & AACG_UNIT_MASK_PRESENT)
                    c *= (meta_l->band[idx] & AACG_META_MS_USED) ? -1 : 1;
                float scale = (float)c * meta_sf(meta_r->band[idx]);
                for (int w = 0; w < info->group_len[g]; w++) {
                    int off = groupOff + w * 128 + offsets[i];
                    int len = offsets[i + 1] - offsets[i];
                    for (int j = 0; j < len; j++)
                        right[off + j] = left[off + j] * scale;
                }
            }
        }
        groupOff += info->group_len[g] * 128;
    }
}

/* tns.js:65-66 */
static const int TNS_MAX_BANDS_1024[13] = {31, 31, 34, 40, 42, 51, 46, 46, 42, 42, 42, 39, 39};
static const int TNS_MAX_BANDS_128[13]  = {9, 9, 10, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14};

/* tns.js:105-177 with the two slips repaired (`ics.maxSFB` -> `info.maxSFB`, `tmp - length` -> `top - length`)
 * and decode = true; see the header for the status of this function. */
int orc_tns_spec(int sample_index, const aacg_chan_info* info, const aacg_tns_info* tns, float* data)
{
    orc_init();
    const int is_short = info->window_sequence == AACG_EIGHT_SHORT_SEQUENCE;
    const uint16_t* swbOffsets = swb_of(sample_index, info);
    const int swbCount = is_short ? g_swb_short_count[sample_index] : g_swb_long_count[sample_index];
    const int maxBands = is_short ? TNS_MAX_BANDS_128[sample_index] : TNS_MAX_BANDS_1024[sample_index];
    const int mmm = maxBands < info->max_sfb ? maxBands : info->max_sfb;          /* tns.js:106 */
    const int windowCount = is_short ? 8 : 1;
    float lpc[20];
    for (int w = 0; w < windowCount; w++) {
        int bottom = swbCount;                                                    /* tns.js:112 */
        for (int filt = 0; filt < tns->n_filt[w]; filt++) {
            const aacg_tns_filter* f = &tns->filt[is_short ? w : filt];
            int top = bottom;
            bottom = top - f->length; if (bottom < 0) bottom = 0;                  /* tns.js:121-123 */
            const int order = f->order;
            if (order == 0) continue;
            if (order > AACG_TNS_MAX_ORDER) return AACG_ERR_UNSUPPORTED;
            for (int i = 0; i < order; i++) {                                     /* tns.js:128-140 */
                const float r = -f->coef[i];
                lpc[i] = r;
                for (int j = 0, len = (i + 1) >> 1; j < len; j++) {
                    const float ff = lpc[j], b = lpc[i - 1 - j];
                    lpc[j] = (float)((double)ff + (double)r * (double)b);
                    lpc[i - 1 - j] = (float)((double)b + (double)r * (double)ff);
                }
            }
            int start = swbOffsets[bottom < mmm ? bottom : mmm];                  /* tns.js:142-146 */
            const int end = swbOffsets[top < mmm ? top : mmm];
            const int size = end - start;
            int inc = 1;
            if (size <= 0) continue;
            if (f->direction) { inc = -1; start = end - 1; }                       /* tns.js:148-151 */
            start += w * 128;
            for (int m = 0; m < size; m++, start += inc)                           /* tns.js:155-163 */
                for (int i = 1; i <= (m < order ? m : order); i++)
                    data[start] = (float)((double)data[start] - (double)data[start - i * inc] * (double)lpc[i - 1]);
        }
    }
    return AACG_OK;
}

/* ics.js:234: randomState = (randomState * (1664525 + 1013904223)) | 0 — the product is
 * formed in double (so it loses low bits) and then wrapped by ToInt32. */
void orc_pns_sequence(int32_t* seq, int n)
{
    int32_t state = 0x1F2E3D4C;                       /* ics.js:31 */
    for (int i = 0; i < n; i++) {
        double p = (double)state * (double)(1664525 + 1013904223);
        double m = fmod(trunc(p), 4294967296.0);       /* ToInt32 */
        if (m < 0) m += 4294967296.0;
        state = (int32_t)(uint32_t)m;
        seq[i] = state;
    }
}

/* ------------------------------------------------------------------------------------ */
/* process(elements) + interleave, decoder.js:201-215, 218-334                            */
/* ------------------------------------------------------------------------------------ */

int orc_decode_batch(int sample_index, int input_kind, int max_streams, int max_channels,
                     const aacg_unit_desc* units, uint32_t n_units,
                     const void* coeffs, const aacg_band_meta* meta,
                     float* pcm_out, float* overlaps, float* spec_out)
{
    return orc_decode_batch_tns(sample_index, input_kind, max_streams, max_channels, units, n_units, coeffs, meta,
                                NULL, AACG_TNS_REFERENCE, pcm_out, overlaps, spec_out);
}

int orc_decode_batch_tns(int sample_index, int input_kind, int max_streams, int max_channels,
                         const aacg_unit_desc* units, uint32_t n_units,
                         const void* coeffs, const aacg_band_meta* meta,
                         const aacg_tns_info* tns, int tns_mode,
                         float* pcm_out, float* overlaps, float* spec_out)
{
    return orc_decode_batch_ex(sample_index, input_kind, max_streams, max_channels, units, n_units, coeffs, meta,
                               tns, tns_mode, AACG_PNS_REFERENCE, pcm_out, overlaps, spec_out);
}

int orc_decode_batch_ex(int sample_index, int input_kind, int max_streams, int max_channels,
                        const aacg_unit_desc* units, uint32_t n_units,
                        const void* coeffs, const aacg_band_meta* meta,
                        const aacg_tns_info* tns, int tns_mode, int pns_mode,
                        float* pcm_out, float* overlaps, float* spec_out)
{
    return orc_decode_batch_cce(sample_index, input_kind, max_streams, max_channels, units, n_units, coeffs, meta,
                                tns, tns_mode, pns_mode, NULL, 0, pcm_out, overlaps, spec_out);
}

/* AACG_CCE_SPEC: cce.js:130-158 (applyDependentCoupling) for one target channel: data += gain[idx] * cce spectrum over
 * the CCE's non-zero bands, with the repairs the code needs to run at all: `swbOffsets[sfb + 1]` for the undefined
 * `swb` (cce.js:149), and the gain list indexed by the band like bandTypes (cce.decode packs it densely, cce.js:85-103,
 * while this loop steps idx with every band).  f32 input has no band types: every band below maxSFB counts. */
static void couple_dependent(int sample_index, const aacg_chan_info* info, const aacg_band_meta* cce_meta,
                             const float* gains, const float* iq, float* data)
{
    const uint16_t* swb = swb_of(sample_index, info);
    int idx = 0, offset = 0;
    for (int g = 0; g < info->group_count; g++) {
        const int len = info->group_len[g];
        for (int sfb = 0; sfb < info->max_sfb; sfb++, idx++) {
            if (cce_meta && meta_bt(cce_meta->band[idx]) == AACG_ZERO_BT) continue;
            const float gain = gains[idx];
            for (int group = 0; group < len; group++)
                for (int k = swb[sfb]; k < swb[sfb + 1]; k++) {
                    const int p = offset + group * 128 + k;
                    data[p] = (float)((double)data[p] + (double)gain * (double)iq[p]);
                }
        }
        offset += len * 128;
    }
}

/* process(elements) + interleave with every optional mode.  cce != NULL (AACG_CCE_SPEC): units flagged AACG_UNIT_CCE
 * are coupling channel elements; decoder.js:406-433 + cce.js:121-158 as they were meant to run —
 *   the `=== isChannelPair` comparison of a number with a boolean (decoder.js:418), the coupling point that becomes 3
 *   instead of 2 (cce.js:69-70) and the loop that stops one coupled element short (decoder.js:416 against cce.js:53) are
 *   the host's business here: it hands over resolved (output channel, gain list) targets;
 *   dependent coupling (points 0, 1) adds the CCE's spectrum band by band before / after the target's TNS;
 *   independent coupling (point 2) adds the CCE's own filterbank output (its own overlap state, at its `channel` beyond
 *   the output channels) times gain[list][0] to the target's — cce.js:121-128 adds the CCE's SPECTRUM to the target's
 *   time signal, which cannot have been the intent.
 * NOT pinned by the reference (it never couples); tests cross-check it against an independent numpy form. */
int orc_decode_batch_cce(int sample_index, int input_kind, int max_streams, int max_channels,
                         const aacg_unit_desc* units, uint32_t n_units,
                         const void* coeffs, const aacg_band_meta* meta,
                         const aacg_tns_info* tns, int tns_mode, int pns_mode,
                         const aacg_cce_info* cce, uint32_t n_cce,
                         float* pcm_out, float* overlaps, float* spec_out)
{
    orc_init();
    if (sample_index < 0 || sample_index > 11) return AACG_ERR_INVALID_ARG;
    enum { MAXU = 32 };
    static __thread float spec[MAXU][2][1024];
    static __thread float cce_time[MAXU][1024];
    float out[1024];

    for (uint32_t n0 = 0; n0 < n_units;) {
        /* one frame: the units that share stream and pcm_offset (include/aacgpu.h) */
        uint32_t n1 = n0 + 1;
        while (n1 < n_units && units[n1].stream == units[n0].stream && units[n1].pcm_offset == units[n0].pcm_offset) n1++;
        if (n1 - n0 > MAXU) return AACG_ERR_CAPACITY;
        const aacg_unit_desc* f = units + n0;
        const int nu = (int)(n1 - n0);
        memset(pcm_out + f->pcm_offset, 0, sizeof(float) * 1024u * f->n_out_ch);   /* decoder.js:229-231 */

        /* spectra of every element, MS / IS inside pairs */
        for (int i = 0; i < nu; i++) {
            const aacg_unit_desc* u = f + i;
            const int is_cce = (u->flags & AACG_UNIT_CCE) != 0;
            if (u->n_ch < 1 || u->n_ch > 2 || (int)u->stream >= max_streams || u->channel + u->n_ch > max_channels ||
                (!is_cce && u->channel + u->n_ch > u->n_out_ch) || (is_cce && (!cce || u->n_ch != 1 || u->reserved1 >= n_cce)))
                return AACG_ERR_INVALID_ARG;
            const aacg_band_meta* ml = NULL; const aacg_band_meta* mr = NULL;
            for (int c = 0; c < u->n_ch; c++) {
                size_t base = ((size_t)u->coef_offset + (size_t)c) * 1024u;
                if (input_kind == AACG_INPUT_SPEC_F32) {
                    memcpy(spec[i][c], (const float*)coeffs + base, sizeof(float) * 1024);
                } else {
                    const aacg_band_meta* m = &meta[u->meta_offset + (uint32_t)c];
                    int rc = orc_dequant_pns(sample_index, &u->ch[c], m, (const int16_t*)coeffs + base, pns_mode, spec[i][c]);
                    if (rc) return rc;
                    if (c == 0) ml = m; else mr = m;
                }
            }
            if (u->n_ch == 2 && input_kind == AACG_INPUT_QUANT_I16) {
                /* processPair, decoder.js:294-302 */
                if ((u->flags & AACG_UNIT_COMMON_WINDOW) && (u->flags & AACG_UNIT_MASK_PRESENT))
                    orc_process_ms(sample_index, u, ml, mr, spec[i][0], spec[i][1]);
                orc_process_is(sample_index, u, ml, mr, spec[i][0], spec[i][1]);
            }
        }
        /* independently switched coupling elements: their own filterbank output */
        for (int i = 0; i < nu; i++) {
            const aacg_unit_desc* u = f + i;
            if (!(u->flags & AACG_UNIT_CCE) || cce[u->reserved1].coupling_point != AACG_CCE_AFTER_IMDCT) continue;
            float* ov = overlaps + ((size_t)u->stream * (size_t)max_channels + (size_t)u->channel) * 1024u;
            orc_filterbank(u->ch[0].window_sequence, u->ch[0].window_shape, u->ch[0].window_shape_prev, spec[i][0], cce_time[i], ov);
        }

        for (int i = 0; i < nu; i++) {
            const aacg_unit_desc* u = f + i;
            if (u->flags & AACG_UNIT_CCE) continue;
            for (int c = 0; c < u->n_ch; c++) {
                const int ch = u->channel + c;
                float* data = spec[i][c];
                /* processSingle / processPair: coupling before TNS, TNS, coupling after TNS (decoder.js:258-266, 304-316).
                 * tns.process is a no-op as the reference runs (tns.js:106,122); AACG_TNS_SPEC: the intended filter */
                for (int point = AACG_CCE_BEFORE_TNS; point <= AACG_CCE_AFTER_TNS; point++) {
                    if (point == AACG_CCE_AFTER_TNS && tns && tns_mode == AACG_TNS_SPEC && (u->ch[c].flags & AACG_CHAN_TNS_PRESENT)) {
                        int rc = orc_tns_spec(sample_index, &u->ch[c], &tns[u->tns_offset + (uint32_t)c], data);
                        if (rc) return rc;
                    }
                    for (int j = 0; j < nu; j++) {
                        const aacg_unit_desc* q = f + j;
                        if (!(q->flags & AACG_UNIT_CCE)) continue;
                        const aacg_cce_info* ci = &cce[q->reserved1];
                        if (ci->coupling_point != point) continue;
                        for (int t = 0; t < ci->n_targets; t++)
                            if (ci->target[t].channel == ch)
                                couple_dependent(sample_index, &q->ch[0], input_kind == AACG_INPUT_QUANT_I16 ? &meta[q->meta_offset] : NULL,
                                                 ci->gain[ci->target[t].gain_list], spec[j][0], data);
                    }
                }
                float* ov = overlaps + ((size_t)u->stream * (size_t)max_channels + (size_t)ch) * 1024u;
                if (spec_out)
                    memcpy(spec_out + ((size_t)u->coef_offset + (size_t)c) * 1024u, data, sizeof(float) * 1024);
                /* filter_bank.process(info, data, this.data[channel], channel), decoder.js:269,318-319 */
                orc_filterbank(u->ch[c].window_sequence, u->ch[c].window_shape, u->ch[c].window_shape_prev, data, out, ov);
                /* coupling after the IMDCT (decoder.js:271-272, 321-322) */
                for (int j = 0; j < nu; j++) {
                    const aacg_unit_desc* q = f + j;
                    if (!(q->flags & AACG_UNIT_CCE)) continue;
                    const aacg_cce_info* ci = &cce[q->reserved1];
                    if (ci->coupling_point != AACG_CCE_AFTER_IMDCT) continue;
                    for (int t = 0; t < ci->n_targets; t++)
                        if (ci->target[t].channel == ch) {
                            const float gain = ci->gain[ci->target[t].gain_list][0];
                            for (int k = 0; k < 1024; k++) out[k] = (float)((double)out[k] + (double)gain * (double)cce_time[j][k]);
                        }
                }
                /* interleave, decoder.js:209-213: output[j++] = data[i][k] / 32768 */
                float* dst = pcm_out + u->pcm_offset + ch;
                for (int k = 0; k < 1024; k++)
                    dst[(size_t)k * u->n_out_ch] = (float)((double)out[k] / 32768.0);
            }
        }
        n0 = n1;
    }
    return AACG_OK;
}
