/*
 * aacgpu.h — C ABI of the MI355X-native AAC-LC synthesis engine.
 *
 * This library replaces ONE seam of audiocogs/aac.js: everything that
 * `AACDecoder.prototype.process(elements)` does plus the channel interleave that
 * follows it in `readChunk` (reference src/decoder.js:201-215, 218-404), i.e.
 *
 *     inverse quantisation      src/ics.js:222-227,244-256  (+ tables src/tables.js:168-191)
 *     mid/side stereo           src/decoder.js:379-404
 *     intensity stereo          src/decoder.js:337-376
 *     TNS                       src/tns.js:105-177   (identity as the reference runs, see AACG_TNS_REFERENCE)
 *     IMDCT + window + OLA      src/filter_bank.js:88-204, src/mdct.js:62-115, src/fft.js:105-192
 *     interleave, /32768        src/decoder.js:203-215
 *
 * The serial part (ADTS demux, raw_data_block parse, Huffman) stays in the JavaScript
 * host, exactly where the reference has it; the host hands batches of parsed
 * elements ("units") across this ABI.  Plain C: pointers and sizes only, int status
 * (0 = OK, <0 = error), no exceptions, no torch/HIP types in any signature.
 *
 * Vocabulary (follows the reference's data model):
 *   channel-frame  one channel x one 1024-sample frame
 *   unit           one syntactic element of one frame: SCE/LFE (1 channel,
 *                  reference `processSingle`, decoder.js:250) or CPE (2 channels,
 *                  `processPair`, decoder.js:285)
 *   stream         one decoder instance's worth of state: the reference keeps
 *                  `FilterBank.overlaps[ch]` (filter_bank.js:38-41) per decoder; the
 *                  engine keeps it per (stream slot, channel) in HBM — in sixteen rotating buffers
 *                  (a launch reads one and writes the next, so that consecutive launches can overlap:
 *                  aacg_decode_pipelined), plus as much again for the hand-over between launches:
 *                  128 KB per (stream slot, channel) in all.
 */
#ifndef AACGPU_H
#define AACGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AACG_ABI_VERSION 6

#define AACG_FRAME_LEN      1024   /* decoder.js:86 frameLength                       */
#define AACG_MAX_SECTIONS   120    /* ics.js:49 MAX_SECTIONS (bandTypes/scaleFactors) */
#define AACG_MAX_CHANNELS   8
#define AACG_RUN_FRAMES     16     /* frames per workgroup run        */

/* ---- status codes ------------------------------------------------------------ */
enum {
    AACG_OK                 = 0,
    AACG_ERR_INVALID_ARG    = -1,
    AACG_ERR_NO_DEVICE      = -2,   /* no HIP device / HIP call failed                 */
    AACG_ERR_OUT_OF_MEMORY  = -3,
    AACG_ERR_CAPACITY       = -4,   /* batch larger than the engine was created for    */
    AACG_ERR_UNSUPPORTED    = -5,   /* reference throws here too (pulse, gain, PNS...) */
    AACG_ERR_LAYOUT_CHANGE  = -6,   /* element layout of a stream changes inside one batch */
    AACG_ERR_STALE_PLAN     = -7,   /* plan does not match the engine's overlap parity */
    AACG_ERR_TIMEOUT        = -8    /* a host wait for the GPU passed the engine's wait limit (aacg_set_wait_limit_ms): the call
                                       returns instead of hanging, aacg_last_error says what was in flight (launch counts, every
                                       stream's and completion event's state, the rendezvous cells' state words).  The work may
                                       still complete later; the engine's state is then undefined — destroy it (aacg_destroy does
                                       not wait again and leaves the engine's device memory allocated)                        */
};

/* ---- window sequences, ics.js:44-47 ------------------------------------------- */
enum {
    AACG_ONLY_LONG_SEQUENCE   = 0,
    AACG_LONG_START_SEQUENCE  = 1,
    AACG_EIGHT_SHORT_SEQUENCE = 2,
    AACG_LONG_STOP_SEQUENCE   = 3
};

/* ---- band types, ics.js:37-42 -------------------------------------------------- */
enum {
    AACG_ZERO_BT       = 0,
    AACG_FIRST_PAIR_BT = 5,
    AACG_ESC_BT        = 11,
    AACG_NOISE_BT      = 13,
    AACG_INTENSITY_BT2 = 14,
    AACG_INTENSITY_BT  = 15
};

/* ---- what the host hands over for the spectrum ------------------------------------ */
enum {
    AACG_INPUT_SPEC_F32  = 0,  /* the spectrum exactly as FilterBank.process receives it
                                  (filter_bank.js:88 `input`): f32[1024] per channel, dequantised,
                                  MS/IS already applied by the host — the filterbank seam       */
    AACG_INPUT_QUANT_I16 = 1   /* the integers Huffman.decodeSpectralData produced
                                  (huffman.js:1462-1490), int16[1024] per channel, same index
                                  order as ICStream.data, + aacg_band_meta per channel — the
                                  process(elements) seam: dequant, MS and IS run on the device   */
};

/* ---- TNS behaviour ------------------------------------------------------------- */
enum {
    AACG_TNS_REFERENCE = 0,    /* what aac.js executes: TNS.process leaves the data untouched
                                  (tns.js:106,122 NaN loop bounds); TNS side
                                  info, if any, is ignored                                       */
    AACG_TNS_SPEC      = 1     /* what tns.js:105-177 was meant to do (and a standard decoder does):
                                  the all-pole filter of its `decode` branch (tns.js:155-163) over the
                                  band range of every filter, per window.  Not aac.js behaviour: an
                                  explicit, separately tested mode                                 */
};

/* ---- PNS behaviour ------------------------------------------------------------- */
enum {
    AACG_PNS_REFERENCE = 0,    /* NOISE_BT bands are refused (AACG_ERR_UNSUPPORTED): aac.js's generator
                                  degenerates to NaN output after 11 draws (ics.js:234,239, SURVEY.md
                                  §8a row 4), there is no reference behaviour to reproduce            */
    AACG_PNS_SPEC      = 1     /* what ics.js:228-243 was meant to do: the generator
                                  state * 1664525 + 1013904223, restarted for every channel of every
                                  frame as the reference's fresh ICStream does, values normalised to the
                                  band's energy scalefactor per window.  Not aac.js behaviour: an explicit,
                                  separately tested mode (QUANT_I16 engines; units flag AACG_UNIT_HAS_PNS) */
};

enum {                          /* aacg_config.cce_mode (ABI version 4)                              */
    AACG_CCE_REFERENCE = 0,     /* coupling channel elements are not part of a batch: aac.js parses them and
                                   never applies them (`1 === true` at decoder.js:418, coupling point 3 at
                                   cce.js:69-70, undefined `swb` at cce.js:149 — SURVEY.md §8a row 9), so the
                                   host drops them; a unit with AACG_UNIT_CCE is refused                   */
    AACG_CCE_SPEC      = 1      /* what cce.js:121-158 + decoder.js:406-433 were written to do: a CCE's own
                                   spectrum, scaled by per-band gains, added to the channels it names —
                                   before or after their TNS (dependent coupling, spectral domain) or, after
                                   its own IMDCT with its own overlap state, to their PCM (independent
                                   coupling).  Not aac.js behaviour: an explicit, separately tested mode
                                   (float32 output engines)                                                */
};

/* ---- coupling side info (AACG_CCE_SPEC), the fields of the reference's CCEElement (cce.js:25-31) after
 * the host has resolved which output channels an element pair / id / select triple means ---- */
#define AACG_CCE_BEFORE_TNS   0      /* CCEElement.BEFORE_TNS  (cce.js:33) */
#define AACG_CCE_AFTER_TNS    1      /* CCEElement.AFTER_TNS   (cce.js:34) */
#define AACG_CCE_AFTER_IMDCT  2      /* CCEElement.AFTER_IMDCT (cce.js:35) */
#define AACG_CCE_MAX_TARGETS  16
typedef struct aacg_cce_target {
    uint8_t channel;           /* output channel of the stream that receives the coupled signal   */
    uint8_t gain_list;         /* which of gain[] applies to it (cce.gain[index], cce.js:77-107)   */
} aacg_cce_target;
typedef struct aacg_cce_info {
    uint8_t coupling_point;    /* AACG_CCE_*                                                       */
    uint8_t n_targets;         /* <= AACG_CCE_MAX_TARGETS                                          */
    uint8_t reserved[2];
    aacg_cce_target target[AACG_CCE_MAX_TARGETS];
    float   gain[AACG_CCE_MAX_TARGETS][AACG_MAX_SECTIONS];   /* dependent: per band, index g * max_sfb + sfb
                                  of the CCE's own ICS; independent: [list][0] for the whole frame      */
} aacg_cce_info;

/* ---- TNS side info (AACG_TNS_SPEC), the fields of the reference's TNS object (tns.js:22-44) ---- */
#define AACG_TNS_MAX_ORDER 12  /* AAC-LC limit; tns.js:84 accepts up to 20, orders 13..20 are refused */
typedef struct aacg_tns_filter {
    uint8_t length;            /* tns.length[w][filt], in scalefactor bands                    */
    uint8_t order;             /* tns.order[w][filt]; 0 = filter absent                        */
    uint8_t direction;         /* tns.direction[w][filt]: 1 = downwards                        */
    uint8_t reserved;
    float   coef[AACG_TNS_MAX_ORDER];   /* tns.coef[w][filt][i] as looked up from TNS_TABLES (tns.js:89-97) */
} aacg_tns_filter;
/* One per channel that has AACG_CHAN_TNS_PRESENT.  Long windows: n_filt[0] <= 3 filters in filt[0..2];
 * EIGHT_SHORT: window w has n_filt[w] <= 1 filter in filt[w] (bit widths of tns.js:47-48).   */
typedef struct aacg_tns_info {
    uint8_t n_filt[8];
    aacg_tns_filter filt[8];
} aacg_tns_info;

/* The ICSInfo fields the path reads (ics.js:270-314), one per channel, 16 bytes. */
typedef struct aacg_chan_info {
    uint8_t window_sequence;    /* info.windowSequence                                         */
    uint8_t window_shape;       /* info.windowShape[1]: 0 sine, 1 KBD                          */
    uint8_t window_shape_prev;  /* info.windowShape[0]; aac.js always has 0 here because it
                                   builds a fresh ICSInfo per frame (decoder.js:145,153)       */
    uint8_t max_sfb;            /* info.maxSFB                                                 */
    uint8_t group_count;        /* info.groupCount (1 for long windows)                        */
    uint8_t flags;              /* AACG_CHAN_*                                                 */
    uint8_t reserved[2];
    uint8_t group_len[8];       /* info.groupLength[g]                                         */
} aacg_chan_info;

#define AACG_CHAN_TNS_PRESENT    0x01   /* ics.tnsPresent (ics.js:71); only read in AACG_TNS_SPEC mode */
#define AACG_UNIT_COMMON_WINDOW 0x01   /* cpe.commonWindow (cpe.js:43)  */
#define AACG_UNIT_MASK_PRESENT  0x02   /* cpe.maskPresent  (cpe.js:47)  */
#define AACG_UNIT_CCE           0x08   /* a coupling channel element (AACG_CCE_SPEC engines): one channel; `channel` is a
                                          stream channel beyond the n_out_ch output channels (it owns overlap state when the
                                          coupling is independent — such an element must then be present in every frame of the
                                          stream's batch, like any element with state — it is never interleaved); reserved1 =
                                          index of its aacg_cce_info                                                     */
#define AACG_UNIT_HAS_PNS       0x04   /* some band of the unit is NOISE_BT (the parser knows, ics.js:84-121):
                                          AACG_PNS_SPEC engines route such batches through the PNS stage,
                                          AACG_PNS_REFERENCE engines refuse them                          */

/* One SCE/LFE/CPE of one frame, 64 bytes.  Units of one stream must be listed in
 * decode order; units that share `stream` and `pcm_offset` form one frame.          */
typedef struct aacg_unit_desc {
    uint32_t stream;        /* stream slot: owner of the overlap state                         */
    uint32_t pcm_offset;    /* float offset of this frame's [1024][n_out_ch] block in pcm_out  */
    uint16_t channel;       /* first output channel of the element (decoder.js:233-247)        */
    uint16_t n_out_ch;      /* channels per frame of this stream = interleave stride
                               (config.chanConfig, decoder.js:219)                             */
    uint8_t  n_ch;          /* 1 = SCE/LFE, 2 = CPE                                            */
    uint8_t  flags;         /* AACG_UNIT_*                                                     */
    uint16_t reserved0;
    uint32_t coef_offset;   /* channel c's spectrum starts at (coef_offset + c) * 1024 elements */
    uint32_t meta_offset;   /* channel c's aacg_band_meta is meta[meta_offset + c] (QUANT only) */
    aacg_chan_info ch[2];   /* [0] = left / the single channel, [1] = right                    */
    uint32_t tns_offset;    /* channel c's aacg_tns_info is tns[tns_offset + c] (if TNS present)  */
    uint32_t reserved1;
} aacg_unit_desc;

/* Per-channel band side info for AACG_INPUT_QUANT_I16, 240 bytes: one 16-bit word per
 * (group, sfb), index g*max_sfb + sfb exactly like ICStream.bandTypes / scaleFactors
 * (ics.js:217).
 *   bits 0..8   index into SCALEFACTOR_TABLE (tables.js:168-176), 0..427
 *   bit  9      negate the looked-up value (noise bands store -SF, ics.js:159)
 *   bit  10     ms_used[idx] (cpe.js:50-63); read from the LEFT channel's meta
 *   bits 12..15 band type (ics.js:37-42)                                            */
typedef struct aacg_band_meta {
    uint16_t band[AACG_MAX_SECTIONS];
} aacg_band_meta;

#define AACG_META_SF_MASK   0x01FFu
#define AACG_META_NEGATE    0x0200u
#define AACG_META_MS_USED   0x0400u
#define AACG_META_BT_SHIFT  12

enum {                          /* aacg_config.output_kind (ABI version 4)                          */
    AACG_OUTPUT_F32 = 0,        /* the reference's output: interleaved float in [-1, 1) (decoder.js:203-215)   */
    AACG_OUTPUT_I16 = 1         /* the same samples as int16: round-to-nearest(x * 32768), saturated — half the
                                   bytes of the path's dominant stream, for hosts that feed 16-bit sinks.  Every
                                   `pcm` pointer of such an engine is an int16_t*, every PCM count is in samples.
                                   Not aac.js behaviour: an explicit output format              */
};

typedef struct aacg_config {
    int32_t abi_version;       /* AACG_ABI_VERSION                                             */
    int32_t device_ordinal;    /* HIP device                                                   */
    int32_t sample_index;      /* config.sampleIndex (decoder.js:63); selects SWB tables, 3 = 48 kHz */
    int32_t max_streams;       /* stream slots with overlap state                              */
    int32_t max_channels;      /* channels per stream, <= AACG_MAX_CHANNELS                    */
    int32_t max_batch_units;   /* capacity of the host-buffer path (aacg_decode_batch)         */
    int32_t input_kind;        /* AACG_INPUT_*                                                 */
    int32_t tns_mode;          /* AACG_TNS_*                                                   */
    int32_t pns_mode;          /* AACG_PNS_* (ABI version 2)                                   */
    int32_t output_kind;       /* AACG_OUTPUT_* (ABI version 4)                                */
    int32_t cce_mode;          /* AACG_CCE_* (ABI version 4)                                   */
} aacg_config;

typedef struct aacg_engine aacg_engine;
typedef struct aacg_plan   aacg_plan;

/* ---- lifetime ------------------------------------------------------------------ */
/* new FilterBank(false, channels) for every stream slot (filter_bank.js:24-44): builds the
 * window / twiddle tables on the device and zeroes all overlap state.                  */
int  aacg_create(const aacg_config* cfg, aacg_engine** out);
void aacg_destroy(aacg_engine* e);
const char* aacg_last_error(const aacg_engine* e);   /* text of the last failure            */
int  aacg_abi_version(void);
/* Every wait of the host for the GPU behind this ABI — aacg_synchronize, aacg_wait, aacg_get/set_overlap, aacg_reset_stream,
 * aacg_plan_destroy, the back-pressure of aacg_decode_pipelined, aacg_decode_batch* — is bounded: after `ms` milliseconds
 * (default 30 000) it returns AACG_ERR_TIMEOUT.  A wait polls for a few microseconds and then sleeps between polls: it does not
 * hold the caller's core.  (`readChunk()` in the reference returns or throws, decoder.js:125-216; it never blocks.)        */
int  aacg_set_wait_limit_ms(aacg_engine* e, uint32_t ms);

/* ---- overlap state (filter_bank.js:38-41) ------------------------------------------ */
int aacg_reset_stream(aacg_engine* e, uint32_t stream);                       /* zero = new FilterBank */
int aacg_get_overlap(aacg_engine* e, uint32_t stream, uint32_t channel, float* dst1024);
int aacg_set_overlap(aacg_engine* e, uint32_t stream, uint32_t channel, const float* src1024);

/* ---- the hot path, host buffers ---------------------------------------------------- */
/* Synchronous equivalent of  process(elements) + interleave  for a whole batch:
 * uploads units/coefficients, runs the kernels, downloads PCM.
 *   coeffs  float[...] (SPEC_F32) or int16_t[...] (QUANT_I16), n_coef_blocks * 1024 elements
 *   meta    aacg_band_meta[n_meta] or NULL (SPEC_F32)
 *   pcm_out float[n_pcm_floats] (int16_t[n_pcm_floats] for AACG_OUTPUT_I16 engines); every frame block [1024][n_out_ch] is fully written
 *           (channels no unit covers are zero, decoder.js:229-231), scale 1/32768.   */
int aacg_decode_batch(aacg_engine* e,
                      const aacg_unit_desc* units, uint32_t n_units,
                      const void* coeffs, uint32_t n_coef_blocks,
                      const aacg_band_meta* meta, uint32_t n_meta,
                      void* pcm_out, size_t n_pcm_floats);

/* Asynchronous pair of the same call, for pipelining: aacg_submit enqueues upload, kernels and
 * download of one batch on one of two internal HIP streams and returns a ticket; aacg_wait blocks
 * until that batch's pcm_out is complete.  Two batches may be in flight: the PCIe transfers of one
 * overlap the kernels of the other, while the kernels themselves run in submission order (they
 * chain through the overlap state).  Buffers must stay valid until the wait; transfers are truly
 * asynchronous only from/to pinned memory (aacg_host_alloc).  Tickets must be waited in order. */
int aacg_submit(aacg_engine* e,
                const aacg_unit_desc* units, uint32_t n_units,
                const void* coeffs, uint32_t n_coef_blocks,
                const aacg_band_meta* meta, uint32_t n_meta,
                void* pcm_out, size_t n_pcm_floats, uint64_t* ticket);
int aacg_wait(aacg_engine* e, uint64_t ticket);
/* The same three calls with TNS side info (AACG_TNS_SPEC engines; tns may be NULL otherwise). */
int aacg_decode_batch_tns(aacg_engine* e,
                          const aacg_unit_desc* units, uint32_t n_units,
                          const void* coeffs, uint32_t n_coef_blocks,
                          const aacg_band_meta* meta, uint32_t n_meta,
                          const aacg_tns_info* tns, uint32_t n_tns,
                          void* pcm_out, size_t n_pcm_floats);
int aacg_submit_tns(aacg_engine* e,
                    const aacg_unit_desc* units, uint32_t n_units,
                    const void* coeffs, uint32_t n_coef_blocks,
                    const aacg_band_meta* meta, uint32_t n_meta,
                    const aacg_tns_info* tns, uint32_t n_tns,
                    void* pcm_out, size_t n_pcm_floats, uint64_t* ticket);
/* ... and with coupling side info as well (AACG_CCE_SPEC engines): every array of a batch in one record. */
typedef struct aacg_batch {
    const aacg_unit_desc* units;  uint32_t n_units;
    const void* coeffs;           uint32_t n_coef_blocks;
    const aacg_band_meta* meta;   uint32_t n_meta;
    const aacg_tns_info* tns;     uint32_t n_tns;      /* NULL / 0: none */
    const aacg_cce_info* cce;     uint32_t n_cce;      /* NULL / 0: none */
    void* pcm_out;                size_t n_pcm_floats;
} aacg_batch;
int aacg_decode_batch_ex(aacg_engine* e, const aacg_batch* b);
int aacg_submit_ex(aacg_engine* e, const aacg_batch* b, uint64_t* ticket);
/* Pinned (page-locked) host memory for the calls above and for aacg_pipeline_*'s PCM.  Blocks under 8 MiB: hipHostMalloc.  Larger
 * ones — a batch's PCM, fresh for every batch of a host that keeps what it was given — are mapped as huge pages, faulted in by a
 * few threads at once and registered with the runtime (32 MiB: 0.6 ms instead of 3.5-6), zero-filled, 2 MiB aligned; both kinds
 * go back through aacg_host_free, from any thread. */
void* aacg_host_alloc(size_t bytes);
void  aacg_host_free(void* p);

/* ---- the hot path, device-resident ------------------------------------------------- */
/* A plan is the uploaded unit table plus the run table the kernel walks.  It can be
 * launched repeatedly: every launch continues the streams where the previous launch of
 * the same plan left them (the next batch of the same shape).                          */
int  aacg_plan_create(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units, aacg_plan** out);
/* with TNS side info (host pointer; becomes part of the plan like the unit table) */
int  aacg_plan_create_tns(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                          const aacg_tns_info* tns, uint32_t n_tns, aacg_plan** out);
int  aacg_plan_create_ex(aacg_engine* e, const aacg_unit_desc* units, uint32_t n_units,
                         const aacg_tns_info* tns, uint32_t n_tns, const aacg_cce_info* cce, uint32_t n_cce, aacg_plan** out);
void aacg_plan_destroy(aacg_plan* p);
/* Launch on `hip_stream` (a hipStream_t passed as void*, NULL = the engine's own stream);
 * returns after enqueueing.  d_coeffs / d_meta / d_pcm are DEVICE pointers.
 * Every call is a NEW launch: the overlap-buffer parity and — for plans whose chains are longer than a run — the epoch of the
 * run-to-run rendezvous are arguments of that launch.  Do not capture a launch in a hipGraph and replay it: call again.
 * Launches of ONE plan are ordered on the device also when they go to different HIP streams (the engine inserts an event
 * wait when the stream changes).  Two PLANS that advance the same audio streams are the caller's to order (same HIP
 * stream, or an event between them): the engine checks their logical order (AACG_ERR_STALE_PLAN), not the device's.      */
int aacg_decode_device(aacg_engine* e, aacg_plan* p,
                       const void* d_coeffs, const aacg_band_meta* d_meta,
                       void* d_pcm, void* hip_stream);
/* Spectral stage only (dequant + MS + IS): writes float[(coef_offset + c) * 1024 ...] so
 * that tests can gate this stage bit-exact.  QUANT_I16 engines only.                   */
int aacg_spectral_device(aacg_engine* e, aacg_plan* p,
                         const void* d_coeffs, const aacg_band_meta* d_meta,
                         float* d_spec_out, void* hip_stream);
/* Waits (on the host) for hip_stream — NULL: the engine's own stream — and for every launch aacg_decode_pipelined has in flight. */
int aacg_synchronize(aacg_engine* e, void* hip_stream);

/* ---- the hot path, device-resident, consecutive launches OVERLAPPED ------------------------------------------------
 * aacg_decode_device puts every launch of a plan behind the one before it: the next batch's first frame needs the last
 * frame's tail (filter_bank.js:38-41, the only state the path carries; filter_bank.js:105-118 is the hand-over).  But that
 * dependency is per (stream, element) chain, not per launch: chain c of launch k + 1 needs chain c's tail of launch k and
 * nothing else.  aacg_decode_pipelined launches on engine-owned HIP streams taken in turn (three, or two for plans whose
 * launches are several rounds of workgroups anyway), so that launch k + 1 starts on the compute units launch k has already
 * left and a unit that is done with launch k + 1 finds work of launch k + 2; consecutive launches' chains meet in rendezvous
 * cells in global memory —
 * whichever side of a chain arrives first publishes what it has (launch k: the windowed tail = the new overlap state;
 * launch k + 1: its windowed first half and where the finished samples go) and leaves, the second finishes the frame.
 * Nobody waits for another workgroup, no dispatch order is assumed, and both arrival orders add the same two rounded
 * numbers: the PCM is bit-identical to aacg_decode_device's.
 *   - Plain batches (float or int16 PCM, no AACG_TNS_SPEC / AACG_PNS_SPEC stage, no coupling element) overlap; every other plan
 *     is accepted and runs behind the launch before it, as aacg_decode_device would run it.
 *   - Inputs must be complete when the call is made, or be produced on a stream the pipeline has been forked from
 *     (aacg_pipeline_fork) since; outputs are complete for work on hip_stream after aacg_pipeline_join(e, hip_stream) and
 *     for the host after aacg_pipeline_join(e, NULL) / aacg_synchronize.
 *   - Launches through other entry points (aacg_decode_device, aacg_submit*, another plan) are ordered behind the
 *     pipeline's by the engine; mixing costs the overlap, never the result.
 *   - BACK-PRESSURE: at most fifteen launches of a sequence are in flight.  A launch reuses the overlap buffers and cells of
 *     the launch sixteen before it, and instead of ordering that on the GPU (a wait in a queue costs what the overlap gains)
 *     the call itself waits — on the host, every second round of launches — until the round four (three streams) or six (two)
 *     rounds back is complete.  A caller that enqueues faster than the GPU decodes is slowed to the GPU's pace; nothing else
 *     changes (aacg_pipeline_order in aacg_routes.cpp is the rule, tests/test_routes.py walks it).
 * Replaces: one `readChunk()` worth of process() + interleave (decoder.js:201-215) per stream and frame, batch after batch. */
int aacg_decode_pipelined(aacg_engine* e, aacg_plan* p, const void* d_coeffs, const aacg_band_meta* d_meta, void* d_pcm);
/* the pipeline's later launches start after everything enqueued on hip_stream so far (inputs produced there) */
int aacg_pipeline_fork(aacg_engine* e, void* hip_stream);
/* work enqueued on hip_stream from now on starts after the pipeline's launches so far; NULL: the host waits for them */
int aacg_pipeline_join(aacg_engine* e, void* hip_stream);
/* What a caller of aacg_decode_pipelined should know about the engine's internal streams: *streams_used = how many of them the
 * current sequence of launches takes in turn (0 before the first pipelined launch); *concurrent = 1 if they were seen to run
 * side by side when the pipeline was set up, 0 if the runtime put them on one hardware queue — pipelined launches are then
 * correct but run one behind the other (the first successful aacg_decode_pipelined also leaves a note in aacg_last_error) —
 * -1 before the pipeline has been set up.  Either pointer may be NULL. */
int aacg_pipeline_info(const aacg_engine* e, int* streams_used, int* concurrent);

/* ---- introspection ---------------------------------------------------------------------------------------------- */
/* Copies the engine's host-built tables (what the device kernels read) for table KATs.
 * which: 0 IQ[8191], 1 SF[428], 2 sine1024, 3 kbd1024, 4 sine128, 5 kbd128               */
int aacg_get_table(aacg_engine* e, int which, float* dst, size_t n);
/* Name of the dominant kernel of the headline route (for matching rocprofv3 rows).      */
const char* aacg_kernel_name(void);
/* The launches aacg_decode_device makes for this plan, by kernel name, " + " between them (dst: n bytes);
 * _ex with pipelined != 0: the launches aacg_decode_pipelined makes. */
int aacg_plan_kernels(aacg_engine* e, const aacg_plan* p, char* dst, size_t n);
int aacg_plan_kernels_ex(aacg_engine* e, const aacg_plan* p, int pipelined, char* dst, size_t n);
/* (measurement and diagnostic entry points — timing marks, the copy calibration, stage-by-stage transforms, hand-made route
 * choices — are declared in aacgpu_tools.h: same library, nothing a host needs) */

/* ---- the bitstream front end on the device (optional; independent of aacg_engine) ---------------
 * One GPU lane parses one frame: what aac.js does serially per frame between `stream.peek(12)` and
 * `this.process(elements)` (decoder.js:126-201 element loop, ics.js:56-201,279-314, cpe.js:37-75,
 * tns.js:68-103, cce.js:45-119 parse-and-discard, huffman.js:1425-1490) for a whole batch of frames at
 * once, writing the records aacg_decode_* / aacg_plan_create* take.  The host keeps what is trivial and
 * serial: finding frame boundaries (ADTS frame_length, MP4 sample sizes) and numbering streams.
 *
 * The parser takes the 12 codebooks as (length, code word, values) lists — book 0 = scalefactors
 * (v[0] = 0..120, huffman.js HCB_SF), books 1..11 = spectral (huffman.js HCB1..HCB11; v[0..3] for books 1-4,
 * v[0..1] for 5-11; magnitudes for the unsigned books 3,4,7..11).  aacg_standard_codebooks() supplies the
 * ones of ISO/IEC 14496-3 (tables 4.A.1-4.A.12), which is what every AAC stream uses; a caller may pass any
 * other complete prefix codes over the same alphabets (the tests do).                                   */
typedef struct aacg_code_entry {
    uint32_t code;             /* the code word, right-aligned                                     */
    uint8_t  len;              /* its length in bits, 1..24                                        */
    int8_t   v[4];
    uint8_t  reserved[3];
} aacg_code_entry;

#define AACG_STANDARD_CODEBOOK_ENTRIES 1362   /* 121 + 4*81 + 2*81 + 2*64 + 2*169 + 289 */
/* Fills entries[AACG_STANDARD_CODEBOOK_ENTRIES] (book after book) and counts[12]; either may be NULL.
 * Returns the number of entries.  Replaces the private tables of src/huffman.js:22-1418.               */
uint32_t aacg_standard_codebooks(aacg_code_entry* entries, uint32_t counts[12]);

typedef struct aacg_parse_frame {
    uint32_t byte_offset;      /* of the frame in `bytes`: an ADTS frame (header included, decoder.js:129-130)
                                  or a bare raw_data_block                                         */
    uint32_t byte_length;
} aacg_parse_frame;

enum {                         /* aacg_parse_result.status; the reference's message for each in
                                  aacg_parse_status_string()                                       */
    AACG_PARSE_OK = 0,
    AACG_PARSE_INSUFFICIENT_DATA = 1,  /* the frame ended inside an element (AV.Bitstream underflow)   */
    AACG_PARSE_BAND_TYPE = 2,          /* ics.js:96  'Invalid band type: 12'                            */
    AACG_PARSE_TOO_MANY_BANDS = 3,     /* ics.js:105 'Too many bands'                                   */
    AACG_PARSE_SCALEFACTOR = 4,        /* ics.js:165 'Scalefactor out of range' (also below -100, where
                                          the reference reads outside its table)                        */
    AACG_PARSE_PULSE_IN_SHORT = 5,     /* ics.js:65  'Pulse tool not allowed in eight short sequence.'  */
    AACG_PARSE_PULSE_RANGE = 6,        /* ics.js:180,193,198 'Pulse SWB / offset out of range'          */
    AACG_PARSE_PULSE_DATA = 7,         /* ics.js:264 'TODO: add pulse data' (without AACG_PARSE_APPLY_PULSES) */
    AACG_PARSE_TNS_ORDER = 8,          /* tns.js:85  'TNS filter out of range' (> 20; > 12 when TNS records
                                          are requested, AACG_TNS_MAX_ORDER)                            */
    AACG_PARSE_PREDICTION = 9,         /* ics.js:318 'Prediction not implemented.'                      */
    AACG_PARSE_GAIN_CONTROL = 10,      /* ics.js:76  'TODO: decode gain control/SSR'                    */
    AACG_PARSE_PCE = 11,               /* decoder.js:184 'TODO: PCE_ELEMENT'                            */
    AACG_PARSE_MAX_SFB = 12,           /* max_sfb beyond the sampling rate's band count (the reference reads
                                          outside its offset table)                                     */
    AACG_PARSE_MS_MASK = 13,           /* cpe.js:66  'Reserved ms mask type: 3'                         */
    AACG_PARSE_ESCAPE = 14,            /* escape sequence longer than the standard's 13 bits            */
    AACG_PARSE_CAPACITY = 15,          /* more elements / channels in the frame than the caller allowed */
    AACG_PARSE_LAYOUT = 16             /* never written by the parser: aacg_plan_refresh_from_parse_ex marks a frame that parsed
                                          well but does not have the elements its stream's plan was made for (the reference decodes
                                          whatever elements a frame brings, decoder.js:233-247; a kept plan cannot)              */
};
#define AACG_PARSE_APPLY_PULSES      0x1u   /* add pulse data to the spectrum (ISO/IEC 14496-3 4.6.3.3) instead
                                               of failing the frame as the reference does              */
#define AACG_PARSE_REFERENCE_QUIRKS  0x2u   /* coupling channel elements consume the bits cce.js consumes
                                               (AFTER_IMDCT never matches, the band index only steps on coded
                                               bands) rather than the standard's syntax                  */
#define AACG_PARSE_SKIP_ZERO_FILL    0x4u   /* aacg_parse_device: do not zero d_q first; positions outside the coded bands then
                                               hold whatever was there — the transform never reads them (bands of type ZERO /
                                               NOISE / INTENSITY and bands beyond max_sfb are not taken from the spectrum) */
#define AACG_PARSE_HAS_PNS 0x1
#define AACG_PARSE_HAS_TNS 0x2
#define AACG_PARSE_HAS_CCE 0x4   /* the frame held a coupling channel element: parsed and dropped (what aac.js executes); a host that
                                    applies coupling (AACG_CCE_SPEC) takes such a frame's records from a front end that keeps them */
typedef struct aacg_parse_result {
    uint8_t  status;           /* AACG_PARSE_*; a failed frame's output records are unspecified
                                  (partially written; never uninitialised memory)                  */
    uint8_t  n_units;          /* SCE/LFE/CPE elements found = records written for this frame      */
    uint8_t  n_channels;
    uint8_t  flags;            /* AACG_PARSE_HAS_*                                                 */
    uint32_t bits_used;        /* consumed from byte_offset on, byte-aligned at the end            */
} aacg_parse_result;

typedef struct aacg_parser aacg_parser;
/* counts[b] entries of book b, concatenated in `entries`; every book must be a complete prefix code.  */
int aacg_parser_create(int device_ordinal, int sample_index, const aacg_code_entry* entries,
                       const uint32_t counts[12], aacg_parser** out);
void aacg_parser_destroy(aacg_parser* p);
const char* aacg_parser_last_error(const aacg_parser* p);
const char* aacg_parse_status_string(int status);
/* aacg_parse_batch's waits for the GPU are bounded like the engine's (aacg_set_wait_limit_ms): AACG_ERR_TIMEOUT after `ms`
 * milliseconds, default 30 000. */
int aacg_parser_set_wait_limit_ms(aacg_parser* p, uint32_t ms);

/* Frame f's element e lands in units[f * max_units + e]; its channels in q / meta / tns block
 * f * max_channels + (running channel index), which is what the record's coef_offset / meta_offset /
 * tns_offset say.  Left for the caller to fill before decoding: stream, pcm_offset, n_out_ch
 * (window_shape_prev is written as 0, as aac.js has it).  reserved0 carries (element type << 4) | id.
 * tns may be NULL: TNS side info is then consumed and dropped (AACG_TNS_REFERENCE engines).
 *
 * aacg_parse_batch: host pointers, returns when the results are there.
 * aacg_parse_device: DEVICE pointers, asynchronous on hip_stream; d_bytes 16-byte aligned with >= 32 readable
 * bytes after the last frame; zero-fills d_units, d_q, d_meta (and d_tns) itself, so a refused frame's records and
 * the element slots beyond a frame's count read as zero.  A parser may be used from several streams in turn: a launch
 * on another stream waits (on the device) for the parser's previous launch, whose lane-order scratch it reuses.   */
int aacg_parse_batch(aacg_parser* p, const uint8_t* bytes, size_t n_bytes,
                     const aacg_parse_frame* frames, uint32_t n_frames,
                     uint32_t max_units, uint32_t max_channels, uint32_t options,
                     aacg_unit_desc* units, int16_t* q, aacg_band_meta* meta, aacg_tns_info* tns,
                     aacg_parse_result* results);
int aacg_parse_device(aacg_parser* p, const void* d_bytes, const aacg_parse_frame* d_frames, uint32_t n_frames,
                      uint32_t max_units, uint32_t max_channels, uint32_t options,
                      aacg_unit_desc* d_units, int16_t* d_q, aacg_band_meta* d_meta, aacg_tns_info* d_tns,
                      aacg_parse_result* d_results, void* hip_stream);
const char* aacg_parse_kernel_name(void);

/* Parser -> transform without the host in between.  A plan's run tables depend on which streams bring how many frames
 * of which element layout, not on what the frames contain; batch after batch of the same streams can therefore keep
 * ONE plan (built once from unit records that carry the structure: stream, pcm_offset, channel, n_out_ch, n_ch and the
 * coef / meta block offsets aacg_parse_device will use: block = frame * max_channels + channel) and
 * have its device unit records rewritten from aacg_parse_device's output: window info and flags
 * come from d_parsed_units (plan unit i <- parsed record i, so the plan's units must be listed frame by frame
 * with max_units per frame), the rest stays.  A frame the parser refused, an element that is not the one the plan
 * expects (other channels, other blocks), or a unit with noise bands (their stage is chosen when a plan is built) becomes a silent unit and is counted in *d_refused (device
 * counter, caller zeroes it).  Asynchronous on hip_stream; follow with aacg_decode_device on the same stream.
 * QUANT_I16 engines with AACG_TNS_REFERENCE only (TNS records are prepared on the host).                      */
int aacg_plan_refresh_from_parse(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* d_parsed_units,
                                 const aacg_parse_result* d_results, uint32_t max_units, uint32_t* d_refused,
                                 void* hip_stream);
/* The same for plans that (a) list only the elements their streams have and (b) are refreshed while earlier launches of theirs
 * are still in flight — what a host does that keeps several batches of the same streams between bytes and PCM at a time.
 *   d_map (device, one per plan unit, or NULL = plan unit i <- parsed record i): where the parser put the unit's element
 *     (frame * max_units + element) and how many elements its frame must have.  The units of a frame must be listed next to
 *     each other in the frame's order.  A frame is then taken or refused AS A WHOLE: wrong element count, an element of other
 *     channels or blocks, a parse error -> every unit of it silent, *d_refused += 1 per frame, and d_results[frame].status
 *     becomes AACG_PARSE_LAYOUT where the parser had said OK (d_results is written).
 *   set: which of the plan's sets of unit records to rewrite (aacg_plan_set_unit_sets); the plan's next launch reads that set.
 *     With more than one set the call does not wait for the plan's launches in flight — the CALLER guarantees that no launch
 *     still reads `set` (e.g. its previous user's output has been waited for on hip_stream) — and a sequence of
 *     aacg_decode_pipelined launches is not interrupted by it: consecutive batches still overlap through the rendezvous cells. */
typedef struct aacg_refresh_map {
    uint32_t parsed_index;     /* frame * max_units + element                                                     */
    uint32_t frame_units;      /* bits 0..7: elements (SCE / CPE / LFE) the frame must have; 0: not checked, refusal per unit.
                                  bits 8..15: how many of them (the first ones) the plan lists, 0 = all — the reference drops
                                  the elements beyond chanConfig channels (decoder.js:233)                               */
} aacg_refresh_map;
int aacg_plan_set_unit_sets(aacg_engine* e, aacg_plan* p, uint32_t n_sets);     /* 1..8, before the plan's first launch */
int aacg_plan_refresh_from_parse_ex(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* d_parsed_units,
                                    aacg_parse_result* d_results, uint32_t max_units, const aacg_refresh_map* d_map,
                                    uint32_t set, uint32_t* d_refused, void* hip_stream);

/* The host's counterpart of aacg_plan_refresh_from_parse, for callers that parse on the CPU (the JavaScript front end) but
 * keep spectra and PCM on the device: batch after batch of the same streams keeps ONE plan, and the next batch's unit
 * records — same streams, frames, elements and PCM positions; new window sequences, shapes, grouping, flags and block
 * offsets — replace the plan's (validated like aacg_plan_create validates them, then one asynchronous copy on hip_stream;
 * follow with aacg_decode_device on the same stream).  AACG_ERR_LAYOUT_CHANGE if the structure differs (nothing is
 * changed then: build a new plan); plans with TNS records, noise bands or coupling elements are built per batch
 * (AACG_ERR_UNSUPPORTED).  Replaces decoder.js:138-198's per-frame `new ICStream` bookkeeping for a whole batch. */
int aacg_plan_refresh_units(aacg_engine* e, aacg_plan* p, const aacg_unit_desc* units, uint32_t n_units, void* hip_stream);

/* ---- bytes in, PCM out: front end and transform resident, ONE call per batch, several batches in flight ---------------
 * What a host of the reference does per stream and per frame in readChunk() (decoder.js:125-216: parse the raw_data_block,
 * process(elements), interleave) for a batch of streams at once: the frames' bytes go up, aacg_parse_device writes the
 * records in HBM, aacg_plan_refresh_from_parse_ex turns them into a kept plan's unit records, aacg_decode_pipelined runs the
 * transform, the PCM comes down; nothing but the frame boundaries (ADTS frame_length) and the stream slots is the host's.
 * A batch takes one of `lanes` sets of device buffers and a HIP stream of its own: batch k's PCM is on its way down PCIe
 * while batch k + 1's bytes go up and are parsed, and the transform launches of consecutive batches of one shape are
 * consecutive launches of ONE plan through aacg_decode_pipelined (they overlap through the rendezvous cells).
 * channels 1 / 2: every frame one SCE / one CPE (channel_configuration 1 / 2; any other frame is refused).  channels 3..8: the
 * reference's chanConfig (decoder.js:77,219: the number of output channels); a stream's element layout — the SCE / CPE / LFE
 * elements of a frame in order, channels dealt out in element order, elements beyond `channels` dropped (decoder.js:233-247) —
 * is learnt from the first frame it submits after aacg_pipeline_reset_stream, and a later frame with other elements is
 * refused as a whole (AACG_PARSE_LAYOUT).  Coupling channel elements are parsed and dropped, as the reference executes them.
 * A pipeline owns an engine (AACG_INPUT_QUANT_I16, AACG_TNS_REFERENCE) and a parser; it is not re-entrant. */
typedef struct aacg_pipeline aacg_pipeline;
typedef struct aacg_pipeline_config {
    int32_t abi_version;       /* AACG_ABI_VERSION                                                          */
    int32_t device_ordinal;
    int32_t sample_index;      /* config.sampleIndex (decoder.js:63)                                        */
    int32_t max_streams;       /* stream slots with overlap state                                           */
    int32_t channels;          /* 1..8: channels of a frame's PCM (the reference's chanConfig)               */
    int32_t max_frames;        /* frames per stream in one batch, at most                                   */
    int32_t output_kind;       /* AACG_OUTPUT_*                                                             */
    int32_t parse_options;     /* AACG_PARSE_* (AACG_PARSE_REFERENCE_QUIRKS for what aac.js reads)          */
    int32_t lanes;             /* batches in flight at most, 1..8; 0 = 5 (ABI version 6)                     */
    int32_t reserved[3];       /* zero                                                                      */
} aacg_pipeline_config;
int  aacg_pipeline_create(const aacg_pipeline_config* cfg, const aacg_code_entry* entries, const uint32_t counts[12], aacg_pipeline** out);
void aacg_pipeline_destroy(aacg_pipeline* p);
const char* aacg_pipeline_last_error(const aacg_pipeline* p);
int  aacg_pipeline_reset_stream(aacg_pipeline* p, uint32_t slot);                 /* new FilterBank for that slot */
/* One batch, synchronous: n_streams streams (slots[s]: the stream slot that owns stream s's overlap state; each slot at most
 * once per batch: AACG_ERR_INVALID_ARG otherwise), the next
 * frames_per_stream frames of each; frames[s * frames_per_stream + f] = frame f of stream s in `bytes` (an ADTS frame,
 * header included, or a bare raw_data_block).  pcm_out: [stream][frame][1024][channels] (float, or int16 for
 * AACG_OUTPUT_I16 pipelines), any host memory — page-locked memory (aacg_host_alloc) receives the PCM straight from the
 * device, other memory through the pipeline's staging and one more host copy.  results (optional): one aacg_parse_result per frame; a frame the parser
 * refused (or whose element is not the one the pipeline was made for) is decoded as silence — its stream's state moves on
 * through a silent frame — and counted in *n_refused: the caller raises the reference's error for it
 * (aacg_parse_status_string) where the frame is reached. */
int  aacg_pipeline_decode(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames,
                          const uint32_t* slots, uint32_t n_streams, uint32_t frames_per_stream,
                          void* pcm_out, aacg_parse_result* results, uint32_t* n_refused);
/* The asynchronous pair of the same call.  aacg_pipeline_submit stages the bytes (they may be reused when it returns),
 * enqueues the batch on the next lane and returns a ticket; pcm_out / results / n_refused are written by the time
 * aacg_pipeline_collect(ticket) returns and must stay valid until then.  Batches decode in submission order (consecutive
 * batches of a stream chain through its overlap state); up to `lanes` may be in flight — submitting one more first finishes
 * the oldest (its outputs are then complete, collecting it later returns at once).  Tickets count from 1.
 * aacg_pipeline_collect waits at most the wait limit (aacg_pipeline_set_wait_limit_ms, default 30 s): AACG_ERR_TIMEOUT, with
 * the lanes' and the engine's state in aacg_pipeline_last_error.  Replaces decoder.js:125-216 for a batch, without the
 * caller's thread waiting for PCIe. */
int  aacg_pipeline_submit(aacg_pipeline* p, const uint8_t* bytes, size_t n_bytes, const aacg_parse_frame* frames,
                          const uint32_t* slots, uint32_t n_streams, uint32_t frames_per_stream,
                          void* pcm_out, aacg_parse_result* results, uint32_t* n_refused, uint64_t* ticket);
int  aacg_pipeline_collect(aacg_pipeline* p, uint64_t ticket);
int  aacg_pipeline_set_wait_limit_ms(aacg_pipeline* p, uint32_t ms);
/* The element layout learnt for a stream slot: returns the number of SCE / CPE / LFE elements of its frames (0: not learnt
 * yet), their channel counts in element_channels[0..7] and how many of them (the first ones) fit `channels` in *kept. */
int  aacg_pipeline_stream_layout(aacg_pipeline* p, uint32_t slot, uint8_t element_channels[8], uint32_t* kept);

#ifdef __cplusplus
}
#endif
#endif /* AACGPU_H */
