/*
 * aacg_tables.cpp — host-side construction of the constant tables the kernels read.
 * Pure C++ (no HIP).  Definitions follow the reference's table generators:
 *   IQ / scalefactor      src/tables.js:168-191
 *   sine / KBD windows    src/filter_bank.js:46-86
 *   MDCT rotation         src/mdct_tables.js (sqrt(2/N) * (cos, sin)(2*pi*(k + 1/8)/N))
 *   SWB offsets           src/tables.js:34-163 (ISO/IEC 14496-3 Tables 4.110-4.128)
 * The FFT inter-stage twiddles are exact roots of unity (the reference's float32
 * recurrence, src/fft.js:59-103, drifts by <= 8.8e-7; parity budget is 1e-4 RMS).
 */
#include "aacg_host.h"

#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>
#include <vector>

namespace {

const double kPi = 3.14159265358979323846;

/* (band width, repeat) runs per sampleIndex; 0 terminates.  Long windows. */
const unsigned char kSwbLong[12][32] = {
    {4,14, 8,5, 12,5, 16,2, 24,1, 28,1, 36,1, 44,1, 64,11, 0},
    {4,14, 8,5, 12,5, 16,2, 24,1, 28,1, 36,1, 44,1, 64,11, 0},
    {4,14, 8,4, 12,3, 16,3, 20,1, 24,2, 28,1, 36,1, 40,18, 0},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,19, 96,1, 0},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,19, 96,1, 0},
    {4,10, 8,7, 12,4, 16,2, 20,2, 24,2, 28,2, 32,22, 0},
    {4,11, 8,10, 12,4, 16,3, 20,2, 24,2, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,2, 64,5, 0},
    {4,11, 8,10, 12,4, 16,3, 20,2, 24,2, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,2, 64,5, 0},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, 0},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, 0},
    {8,11, 12,9, 16,4, 20,3, 24,2, 28,2, 32,1, 36,1, 40,2, 44,1, 48,1, 52,1, 56,1, 60,1, 64,3, 0},
    {12,13, 16,7, 20,4, 24,3, 28,2, 32,1, 36,2, 40,1, 44,1, 48,1, 52,1, 56,1, 60,1, 64,1, 80,1, 0},
};
/* Short windows. */
const unsigned char kSwbShort[12][12] = {
    {4,6, 8,3, 16,1, 28,1, 36,1, 0},
    {4,6, 8,3, 16,1, 28,1, 36,1, 0},
    {4,6, 8,3, 16,1, 28,1, 36,1, 0},
    {4,5, 8,3, 12,3, 16,3, 0},
    {4,5, 8,3, 12,3, 16,3, 0},
    {4,5, 8,3, 12,3, 16,3, 0},
    {4,7, 8,3, 12,2, 16,2, 20,1, 0},
    {4,7, 8,3, 12,2, 16,2, 20,1, 0},
    {4,8, 8,2, 12,2, 16,1, 20,2, 0},
    {4,8, 8,2, 12,2, 16,1, 20,2, 0},
    {4,8, 8,2, 12,2, 16,1, 20,2, 0},
    {4,7, 8,4, 12,1, 16,1, 20,2, 0},
};

int expand(const unsigned char* rle, std::vector<int>& off)
{
    off.assign(1, 0);
    for (int i = 0; rle[i]; i += 2)
        for (int r = 0; r < rle[i + 1]; r++) off.push_back(off.back() + rle[i]);
    return (int)off.size() - 1;
}

void sine_window(float* d, int len)
{
    for (int i = 0; i < len; i++) d[i] = (float)std::sin((i + 0.5) * (kPi / (2.0 * len)));
}

/* Kaiser-Bessel derived window as filter_bank.js:54-79 computes it: 50-term Horner series for
 * I0, cumulative sums kept in float32 while the running total stays double and gets +1. */
void kbd_window(float* out, double alpha, int len)
{
    const double pin = kPi / len, a2 = (alpha * pin) * (alpha * pin);
    std::vector<float> cum(len);
    double total = 0.0;
    for (int n = 0; n < len; n++) {
        const double t = (double)n * (double)(len - n) * a2;
        double b = 1.0;
        for (int j = 50; j > 0; j--) b = b * t / (double)(j * j) + 1.0;
        total += b;
        cum[n] = (float)total;
    }
    total += 1.0;
    for (int n = 0; n < len; n++) out[n] = (float)std::sqrt((double)cum[n] / total);
}

}  // namespace

int aacg_swb_offsets(int sample_index, int is_long, int* dst /* >= 52 */)
{
    if (sample_index < 0 || sample_index > 11) return 0;
    std::vector<int> off;
    const int n = expand(is_long ? kSwbLong[sample_index] : kSwbShort[sample_index], off);
    for (int i = 0; i <= n; i++) dst[i] = off[i];
    return n;
}

int aacg_build_tables(int sample_index, aacg_tables* t, aacg_host_windows* hw)
{
    if (sample_index < 0 || sample_index > 11) return AACG_ERR_INVALID_ARG;
    std::memset(t, 0, sizeof *t);

    for (int k = 0; k < 512; k++) {                 /* k = l + 64 j stored [j][l] */
        const double a = 2.0 * kPi * (k + 0.125) / 2048.0, s = std::sqrt(2.0 / 2048.0);
        t->sincos_long[k >> 6][k & 63].re = (float)(s * std::cos(a));
        t->sincos_long[k >> 6][k & 63].im = (float)(s * std::sin(a));
    }
    for (int k = 0; k < 64; k++) {                  /* k = g + 8 j stored [j][g] */
        const double a = 2.0 * kPi * (k + 0.125) / 256.0, s = std::sqrt(2.0 / 256.0);
        t->sincos_short[k >> 3][k & 7].re = (float)(s * std::cos(a));
        t->sincos_short[k >> 3][k & 7].im = (float)(s * std::sin(a));
    }
    for (int q = 1; q < 8; q++) {
        for (int l = 0; l < 64; l++) {
            const double a = 2.0 * kPi * (double)(l * q) / 512.0;
            t->tw512[q - 1][l].re = (float)std::cos(a);
            t->tw512[q - 1][l].im = (float)std::sin(a);
        }
        for (int g = 0; g < 8; g++) {
            const double a = 2.0 * kPi * (double)(g * q) / 64.0;
            t->tw64[q - 1][g].re = (float)std::cos(a);
            t->tw64[q - 1][g].im = (float)std::sin(a);
        }
    }

    aacg_host_windows w;
    sine_window(w.sine_long, 1024);
    sine_window(w.sine_short, 128);
    kbd_window(w.kbd_long, 4.0, 1024);       /* filter_bank.js:83 */
    kbd_window(w.kbd_short, 6.0, 128);       /* filter_bank.js:84 */
    if (hw) *hw = w;
    /* The device copies carry the output scale 1/32768 of decoder.js:211 (AACG_PCM_SCALE, a power of two: exact), so that
     * windowed heads and tails are PCM-scaled where they are made and the overlap-add is a bare addition:
     * (ov + x w) / 32768 == ov / 32768 + x (w / 32768) bit for bit.  The overlap state in HBM is at that scale too
     * (aacg_get / set_overlap convert). */
    for (int i = 0; i < 1024; i++) { t->win_long[0][i] = w.sine_long[i] * AACG_PCM_SCALE; t->win_long[1][i] = w.kbd_long[i] * AACG_PCM_SCALE; }
    for (int i = 0; i < 128; i++) { t->win_short[0][i] = w.sine_short[i] * AACG_PCM_SCALE; t->win_short[1][i] = w.kbd_short[i] * AACG_PCM_SCALE; }

    for (int i = 0; i < 8191; i++) t->iq[i] = (float)std::pow((double)i, 4.0 / 3.0);
    t->iq[8191] = std::numeric_limits<float>::quiet_NaN();
    for (int q = -512; q < 512; q++) t->iq_signed[q + 512] = q > 0 ? t->iq[q] : -t->iq[-q];
    for (int i = 0; i < 428; i++) t->sf[i] = (float)std::pow(2.0, (i - 200) / 4.0);

    std::vector<int> off;
    int n = expand(kSwbLong[sample_index], off);
    for (int b = 0; b < n; b++)
        for (int k = off[b]; k < off[b + 1]; k++) t->band_of_long[k] = (uint8_t)b;
    n = expand(kSwbShort[sample_index], off);
    for (int b = 0; b < n; b++)
        for (int k = off[b]; k < off[b + 1]; k++) t->band_of_short[k] = (uint8_t)b;
    static_assert(offsetof(aacg_tables, tw512) == 4 * AACG_TAB_OFF_TW512, "table map");
    static_assert(offsetof(aacg_tables, tw64) == 4 * AACG_TAB_OFF_TW64, "table map");
    static_assert(offsetof(aacg_tables, sincos_short) == 4 * AACG_TAB_OFF_SINCOS_SHORT, "table map");
    static_assert(offsetof(aacg_tables, win_long) == 4 * AACG_TAB_OFF_WIN_LONG, "table map");
    static_assert(offsetof(aacg_tables, win_short) == 4 * AACG_TAB_OFF_WIN_SHORT, "table map");
    static_assert(offsetof(aacg_tables, sf) == 4 * AACG_TAB_OFF_SF, "table map");
    static_assert(offsetof(aacg_tables, iq_signed) == 4 * AACG_TAB_OFF_IQ_SMALL, "table map");
    static_assert(offsetof(aacg_tables, band_of_long) == 4 * AACG_TAB_OFF_BAND_LONG, "table map");
    static_assert(offsetof(aacg_tables, band_of_short) == 4 * AACG_TAB_OFF_BAND_SHORT, "table map");
    static_assert(offsetof(aacg_tables, iq) == 4 * AACG_TAB_QUANT_FLOATS, "table map");
    static_assert(AACG_LDS_BYTES_QUANT <= 160 * 1024, "LDS budget of one CU");
    return AACG_OK;
}

/* AACG_PNS_SPEC: the generator's sequence, the running sum of its squares, the band offsets (ics.js:228-243). */
int aacg_build_pns_tables(int sample_index, aacg_pns_tables* t)
{
    if (sample_index < 0 || sample_index > 11) return AACG_ERR_INVALID_ARG;
    std::memset(t, 0, sizeof *t);
    uint32_t state = 0x1F2E3D4Cu;                       /* ics.js:31 */
    double sum = 0.0;
    for (int p = 0; p < 1024; p++) {
        state = state * 1664525u + 1013904223u;         /* the intended recurrence, wrapping like |0 */
        t->rnd[p] = (float)(int32_t)state;
        t->esum[p] = sum;
        sum += (double)t->rnd[p] * (double)t->rnd[p];
    }
    t->esum[1024] = sum;
    int off[64];
    int n = aacg_swb_offsets(sample_index, 1, off);
    for (int i = 0; i <= n && i < 64; i++) t->swb_long[i] = (uint16_t)off[i];
    n = aacg_swb_offsets(sample_index, 0, off);
    for (int i = 0; i <= n && i < 16; i++) t->swb_short[i] = (uint16_t)off[i];
    return AACG_OK;
}
