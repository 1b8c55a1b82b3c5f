/*
 * aacg_device.h — data the kernels read: constant tables, the run table, launch params.
 * Plain structs shared by the host side (aacg_tables.cpp, aacg_engine.hip) and the kernels.
 */
#ifndef AACG_DEVICE_H
#define AACG_DEVICE_H

#include <stdint.h>
#include "../../include/aacgpu.h"

#define AACG_RUN_W        AACG_RUN_FRAMES          /* frames per run = working waves per workgroup */
#define AACG_WG_WAVES     (AACG_RUN_W + 1)         /* + wave 0: predecessor tail                    */
#define AACG_WG_THREADS   (AACG_WG_WAVES * 64)

/* LDS per wave: tail[0] (1024 f32) + max(tail[1], FFT exchange scratch 576 complex) */
#define AACG_SCRATCH_C2   576                      /* 8 rows x 72 = 64 lanes x 9 (padded)          */
#define AACG_SLOT_FLOATS  (1024 + 2 * AACG_SCRATCH_C2)
#define AACG_SLOT_BYTES   (AACG_SLOT_FLOATS * 4)
#define AACG_WG_LDS_BYTES (AACG_WG_WAVES * AACG_SLOT_BYTES)

struct aacg_c2 { float re, im; };

/* All constant tables, one device allocation (~61 KiB); most waves touch ~12 KiB of it
 * (ONLY_LONG: sincos_long, tw512, tw64, one head window, one tail window), which stays
 * resident in the 32 KiB vector L1 of each CU. */
struct aacg_tables {
    /* mdct.js:73-87 twiddles sqrt(2/N) * (cos, sin)(2*pi*(k + 1/8)/N), f32 */
    aacg_c2 sincos_long[512];
    aacg_c2 sincos_short[64];
    /* radix-8 inter-stage twiddles e^{+2*pi*i*l*q/512}, [q-1][l], and e^{+2*pi*i*g*r/64}, [r-1][g] */
    aacg_c2 tw512[7][64];
    aacg_c2 tw64[7][8];
    /* effective 1024-sample windows of the long sequences (filter_bank.js:105-141,180-202), natural order:
     * head_win[(stop ? 2 : 0) + shape_prev][n]  multiplies IMDCT output n       (first half)
     * tail_win[(start ? 2 : 0) + shape][n]      multiplies IMDCT output 1024+n  (second half) */
    float head_win[4][1024];
    float tail_win[4][1024];
    float short_win[2][128];          /* SINE_128, KBD_128 (filter_bank.js:82,84) */
    float iq[8192];                   /* IQ_TABLE (tables.js:182-191); [8191] = NaN like the JS out-of-range read */
    float sf[512];                    /* SCALEFACTOR_TABLE (tables.js:168-176), 428 used */
    uint8_t band_of_long[1024];       /* coefficient -> sfb for this sample_index (tables.js:34-155) */
    uint8_t band_of_short[128];
};

/* One workgroup's work: up to AACG_RUN_W consecutive frames of one element of one stream. */
struct aacg_run {
    int32_t pred_unit;                /* unit whose tail feeds unit[0]; -1: take it from overlap state */
    int32_t n_units;
    int32_t unit[AACG_RUN_W];
    /* The chain's overlap state is double-buffered (a run that reads it and the run that
     * writes it are different workgroups of one launch): float offsets of the two buffers
     * per channel.  Launch parity `flip` selects in = flip ? b : a, out = flip ? a : b.   */
    int32_t ov_a[2];
    int32_t ov_b[2];
    int32_t is_last;                  /* last run of its chain: the final tail goes to `out` */
    int32_t reserved;
};

struct aacg_kparams {
    const aacg_unit_desc* units;
    const aacg_run*       runs;
    const void*           coeffs;     /* float or int16_t, per input kind */
    const aacg_band_meta* meta;
    float*                pcm;
    float*                overlap;    /* overlap pool */
    float*                spec_out;   /* spectral-only kernel */
    const aacg_tables*    tab;
    int32_t               flip;       /* 0/1: swap ov_in and ov_out (plan reuse, see aacg_engine.hip) */
    int32_t               n_runs;
};

#endif
