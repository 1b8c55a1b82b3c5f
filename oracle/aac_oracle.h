/*
 * aac_oracle.h — CPU restatement of aac.js's per-frame transform path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or
 * executed from the product library (aac.js_amd/); it is the checker that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg compare the HIP path with.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against outputs
 * of the reference itself (audiocogs/aac.js v0.1.3 run under Node 12 in the build
 * container) — tests/golden/golden.bin, produced by tests/golden/gen/gen_golden.js.  The
 * reference ships no tests or golden vectors of its own.
 *
 * Rounding model: JavaScript numbers are binary64; a value becomes
 * binary32 only when stored into a Float32Array.  Compile with -ffp-contract=off.
 */
#ifndef AAC_ORACLE_H
#define AAC_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include "../include/aacgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Build all tables (idempotent).  tables.js:168-191, filter_bank.js:46-86,
 * fft.js:59-103, mdct_tables.js (by formula). */
void orc_init(void);
/* exact = 1: correctly rounded FFT roots instead of the reference's float32 recurrence (fft.js:59-103); 0 restores it.
 * Test aid for attributing the GPU engine's distance from the reference; never changes golden parity. */
void orc_set_fft_roots(int exact);

/* Table access for known-answer tests.
 * which: 0 IQ_TABLE[8191] f32        1 SCALEFACTOR_TABLE[428] f32
 *        2 SINE_1024 f32             3 KBD_1024 f32
 *        4 SINE_128 f32              5 KBD_128 f32
 *        6 FFT(512).roots [512][3] f32   7 FFT(64).roots [64][2] f32
 * returns element count, copies min(n, count) floats. */
size_t orc_get_table_f32(int which, float* dst, size_t n);
/* which: 0 MDCT_TABLE_2048 [512][2] f64   1 MDCT_TABLE_256 [64][2] f64 */
size_t orc_get_table_f64(int which, double* dst, size_t n);
/* SWB offsets (tables.js:34-163): long=1 -> SWB_OFFSET_1024[sample_index], else _128.
 * Writes count+1 offsets, returns count (0 if sample_index is out of range). */
int orc_get_swb_offsets(int sample_index, int is_long, uint16_t* dst);

/* fft.js:105-192 with forward=false; len 512 or 64; buf is len x {re,im}, in place. */
void orc_fft_inverse(int len, float* buf);
/* mdct.js:62-115; N 2048 or 256; in[N/2] -> out[N]. */
void orc_imdct(int N, const float* in, float* out);
/* filter_bank.js:88-204.  overlap[1024] is this channel's state, read and written. */
void orc_filterbank(int window_sequence, int window_shape, int window_shape_prev,
                    const float* in, float* out, float* overlap);

/* ics.js:222-227,244-256 for one channel (no PNS: NOISE_BT returns AACG_ERR_UNSUPPORTED,
 * see SURVEY.md §8a row 4).  q[1024] in ICStream.data index order.  */
int orc_dequant(int sample_index, const aacg_chan_info* info, const aacg_band_meta* meta,
                const int16_t* q, float* data);
/* decoder.js:379-404 and :337-376 on a channel pair, in place. */
void orc_process_ms(int sample_index, const aacg_unit_desc* u,
                    const aacg_band_meta* meta_l, const aacg_band_meta* meta_r,
                    float* left, float* right);
void orc_process_is(int sample_index, const aacg_unit_desc* u,
                    const aacg_band_meta* meta_l, const aacg_band_meta* meta_r,
                    const float* left, float* right);
/* AACG_TNS_SPEC: tns.js:105-177 as it was meant to run (the reference never executes it):
 * `mmm = min(maxBands, info.maxSFB)`, `bottom = max(0, top - length)`, the all-pole `decode` branch (:155-163),
 * float32 stores as in the reference's typed arrays; short windows use TNS_MAX_BANDS_128 (tns.js:66).
 * NOT pinned by the reference (there is no reference behaviour to pin to); tests cross-check it against an
 * independent double-precision direct form. */
int orc_tns_spec(int sample_index, const aacg_chan_info* info, const aacg_tns_info* tns, float* data);

/* PNS generator exactly as written (ics.js:234): fills seq[n] starting from 0x1F2E3D4C. */
void orc_pns_sequence(int32_t* seq, int n);

/* process(elements) + interleave for a batch of units (decoder.js:201-215, 218-334),
 * in unit order.  overlaps is [max_streams][max_channels][1024], read and written.
 * If spec_out != NULL the reconstructed spectrum (after MS/IS) of channel c of a unit is
 * also copied to spec_out[(coef_offset + c) * 1024 ...]. */
int orc_decode_batch(int sample_index, int input_kind, int max_streams, int max_channels,
                     const aacg_unit_desc* units, uint32_t n_units,
                     const void* coeffs, const aacg_band_meta* meta,
                     float* pcm_out, float* overlaps, float* spec_out);
/* dequant with a PNS mode (AACG_PNS_SPEC: noise bands as ics.js:228-243 was meant to fill them; unpinned) */
int orc_dequant_pns(int sample_index, const aacg_chan_info* info, const aacg_band_meta* meta,
                    const int16_t* q, int pns_mode, float* data);
/* all modes */
int orc_decode_batch_ex(int sample_index, int input_kind, int max_streams, int max_channels,
                        const aacg_unit_desc* units, uint32_t n_units,
                        const void* coeffs, const aacg_band_meta* meta,
                        const aacg_tns_info* tns, int tns_mode, int pns_mode,
                        float* pcm_out, float* overlaps, float* spec_out);

/* ... and coupling channel elements (AACG_CCE_SPEC; cce == NULL: none; unpinned, see aac_oracle.c) */
int orc_decode_batch_cce(int sample_index, int input_kind, int max_streams, int max_channels,
                         const aacg_unit_desc* units, uint32_t n_units,
                         const void* coeffs, const aacg_band_meta* meta,
                         const aacg_tns_info* tns, int tns_mode, int pns_mode,
                         const aacg_cce_info* cce, uint32_t n_cce,
                         float* pcm_out, float* overlaps, float* spec_out);

/* the same with TNS side info: tns == NULL or tns_mode == AACG_TNS_REFERENCE leaves the spectrum untouched */
int orc_decode_batch_tns(int sample_index, int input_kind, int max_streams, int max_channels,
                         const aacg_unit_desc* units, uint32_t n_units,
                         const void* coeffs, const aacg_band_meta* meta,
                         const aacg_tns_info* tns, int tns_mode,
                         float* pcm_out, float* overlaps, float* spec_out);

/* orc_bench.c: the CPU baseline's timing loop — n_threads workers, each decoding its own copy of the batch (its own
 * stream set) for ~seconds; returns whole batches decoded (< 0: error), *elapsed = the longest worker's time. */
long long orc_bench_threads(int n_threads, double seconds, int sample_index, int input_kind, int max_streams, int max_channels,
                            const aacg_unit_desc* units, uint32_t n_units, const void* coeffs, size_t coeff_bytes,
                            const aacg_band_meta* meta, size_t n_meta, size_t n_pcm_floats, double* elapsed);

#ifdef __cplusplus
}
#endif
#endif
