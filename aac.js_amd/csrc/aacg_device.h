/*
 * aacg_device.h — data the kernels read: constant tables, the run table, launch params.
 * Plain structs shared by the host side (aacg_tables.cpp, aacg_engine.hip) and the kernels.
 */
#ifndef AACG_DEVICE_H
#define AACG_DEVICE_H

#include <stdint.h>
#include "../../include/aacgpu.h"
#include "../../include/aacgpu_tools.h"

#define AACG_RUN_W        AACG_RUN_FRAMES          /* waves per workgroup = frames per first run of a chain */
#define AACG_WG_WAVES     AACG_RUN_W
#define AACG_WG_THREADS   (AACG_WG_WAVES * 64)

/* decoder.js:211: PCM = sample / 32768.  Folded into the device window tables (aacg_tables.cpp). */
#define AACG_PCM_SCALE (1.0f / 32768.0f)

struct aacg_c2 { float re, im; };

/* ---- constant tables ---------------------------------------------------------------- */
/* One device allocation; every workgroup copies the part its kernel needs into LDS once
 * (16-byte loads, L2-resident) and all 16 waves read it from there: the per-wave table
 * traffic never touches the vector-memory pipeline.  Offsets in floats. */
struct aacg_tables {
    /* mdct.js:73-87 rotation sqrt(2/N) * (cos, sin)(2*pi*(k + 1/8)/N), k = l + 64 j stored [j][l] */
    aacg_c2 sincos_long[8][64];
    /* radix-8 inter-stage twiddles e^{+2*pi*i*l*q/512}, [q-1][l], and e^{+2*pi*i*g*r/64}, [r-1][g] */
    aacg_c2 tw512[7][64];
    aacg_c2 tw64[7][8];
    aacg_c2 sincos_short[8][8];       /* N = 256, k = g + 8 j stored [j][g] */
    float win_long[2][1024];          /* SINE_1024, KBD_1024 (filter_bank.js:81,83), times AACG_PCM_SCALE */
    float win_short[2][128];          /* SINE_128,  KBD_128  (filter_bank.js:82,84), times AACG_PCM_SCALE */
    /* ---- end of the part the f32 kernel stages (AACG_TAB_F32_FLOATS) ---- */
    float sf[432];                    /* SCALEFACTOR_TABLE (tables.js:168-176), 428 used */
    float iq_signed[1024];            /* sign(q) * IQ_TABLE[|q|] for q = -512..511 at [q + 512] (tables.js:182-191, ics.js:251:
                                         q = 0 gives -0); larger magnitudes take the full table below */
    uint8_t band_of_long[1024];       /* coefficient -> sfb for this sample_index (tables.js:34-155) */
    uint8_t band_of_short[128];
    /* ---- end of the part the quant kernel stages (AACG_TAB_QUANT_FLOATS) ---- */
    float iq[8192];                   /* full IQ_TABLE, global; [8191] = NaN like the JS out-of-range read */
};

#define AACG_TAB_OFF_SINCOS_LONG  0
#define AACG_TAB_OFF_TW512        (AACG_TAB_OFF_SINCOS_LONG + 8 * 64 * 2)
#define AACG_TAB_OFF_TW64         (AACG_TAB_OFF_TW512 + 7 * 64 * 2)
#define AACG_TAB_OFF_SINCOS_SHORT (AACG_TAB_OFF_TW64 + 7 * 8 * 2)
#define AACG_TAB_OFF_WIN_LONG     (AACG_TAB_OFF_SINCOS_SHORT + 8 * 8 * 2)
#define AACG_TAB_OFF_WIN_SHORT    (AACG_TAB_OFF_WIN_LONG + 2 * 1024)
#define AACG_TAB_F32_FLOATS       (AACG_TAB_OFF_WIN_SHORT + 2 * 128)
#define AACG_TAB_OFF_SF           AACG_TAB_F32_FLOATS
#define AACG_TAB_OFF_IQ_SMALL     (AACG_TAB_OFF_SF + 432)
#define AACG_TAB_OFF_BAND_LONG    (AACG_TAB_OFF_IQ_SMALL + 1024)
#define AACG_TAB_OFF_BAND_SHORT   (AACG_TAB_OFF_BAND_LONG + 256)
#define AACG_TAB_QUANT_FLOATS     (AACG_TAB_OFF_BAND_SHORT + 32)

/* ---- LDS map of the run kernel -------------------------------------------------------- */
/* [tables][slot 0] ... [slot 15]; slot = two 1024-float areas, one per channel.  An area
 * holds, in turn, the channel's staged spectrum (natural order), its FFT transposes
 * (512 complex, XOR-swizzled instead of padded) and finally its windowed tail, which the
 * next wave reads after the workgroup barrier. */
#define AACG_SLOT_FLOATS  2048
/* the slots start on a 512-byte boundary: the kernels XOR small offsets into slot addresses (long_pair), and the
 * strided two-address LDS reads (ds_read2st64_b64) count their offsets in units of 512 bytes */
#define AACG_TAB_SLOT_BASE(tab_floats) (((tab_floats) + 127) & ~127)
#define AACG_LDS_FLOATS(tab_floats) (AACG_TAB_SLOT_BASE(tab_floats) + AACG_WG_WAVES * AACG_SLOT_FLOATS + AACG_WG_WAVES)   /* + one hand-off flag per wave */
#define AACG_LDS_BYTES_F32    (4 * AACG_LDS_FLOATS(AACG_TAB_F32_FLOATS))
#define AACG_LDS_BYTES_QUANT  (4 * AACG_LDS_FLOATS(AACG_TAB_QUANT_FLOATS))

/* Device copy of a unit: the ABI record plus what the planner derives from it. */
struct aacg_dev_unit {
    aacg_unit_desc d;
    uint32_t gmap[2];                 /* per channel: group of window w in bits 4w..4w+3 (ics.js:288-296 grouping) */
    /* AACG_CCE_SPEC, independent coupling applied where the target's PCM is formed: this unit's jobs in the plan's
     * fused-job list (aacg_couple_job: src = block of the side buffer, dst = channel 0 / 1 of this unit), in the order
     * of the frame's coupling elements; 0 jobs otherwise */
    uint32_t cpl_first, cpl_n;
};

/* Device form of one channel's TNS side info (AACG_TNS_SPEC): per filter slot the sample range and the
 * direct-form coefficients, both derived on the host exactly as tns.js:111-152 derives them.  Long windows
 * use slots 0..2, EIGHT_SHORT uses slot w for window w; order 0 = empty slot. */
struct aacg_dev_tns {
    int32_t start[8];                 /* first sample filtered, absolute index in ICStream.data order (includes w * 128) */
    int32_t size[8];                  /* number of samples; multiple of 4 */
    int32_t inc[8];                   /* +1 upwards, -1 downwards */
    int32_t order[8];
    float   lpc[8][AACG_TNS_MAX_ORDER];
};

/* AACG_PNS_SPEC: the noise generator's whole sequence (it restarts for every channel of every frame, so it is a
 * fixed table), the running sum of its squares (a band-window's energy is a difference of two entries) and the
 * scalefactor-band offsets of the engine's sample rate.  Global memory, read only where a band is NOISE_BT. */
struct aacg_pns_tables {
    float    rnd[1024];               /* rnd[p] = (float)state after p + 1 steps of state * 1664525 + 1013904223 from 0x1F2E3D4C */
    double   esum[1028];              /* esum[p] = sum of rnd[i]^2 for i < p, p = 0..1024 */
    uint16_t swb_long[64];            /* swbOffsets of the long / short window (tables.js:34-155), [swbCount] = 1024 / 128 */
    uint16_t swb_short[16];
};

/* The overlap state (filter_bank.js:38-41) of one channel lives in AACG_OV_BUFFERS rotating buffers of 1024 floats: a launch
 * that advances the channel reads buffer r and leaves the new state in buffer r + 1 (mod AACG_OV_BUFFERS).  Two would do for
 * launches that follow each other on one HIP stream; more are what lets consecutive launches of a plan OVERLAP
 * (aacg_decode_pipelined): launch n + 1 may be writing buffer r + 2 while launch n still reads r and writes r + 1.
 * A buffer (and the rendezvous cell that goes with it) comes round again after AACG_OV_BUFFERS launches: launch n shares
 * buffers with launches n - K + 1, n - K and n - K - 1 (K = AACG_OV_BUFFERS), so it must start behind EVERY launch up to
 * n - K + 1.  The engine issues launch n on stream n mod AACG_PIPE_STREAMS (behind n - 3, n - 6, ... and nothing else on the
 * GPU's side) and bounds the rest from the HOST (back-pressure): the launches of every second round of three carry a
 * completion event, and before the first launch of such a round is enqueued the host waits for the events of the round FOUR
 * back (aacg_pipeline_order, aacg_routes.cpp).  Everything up to launch 3 (r - 4) + 2 is then complete before launches 3 r ..
 * 3 r + 5 exist: a distance of at most fifteen, hence sixteen buffers, and the GPU's queues hold three rounds (three launches
 * running, six waiting) when the host comes back to enqueue two more.  What this replaced, measured on the way
 * (gpurun_out/ab_*.txt of round 5, profiles/r05_budget.txt): two streams and five buffers with an event WAIT per four
 * launches kept a CU waiting 1.6-2.5 us between two workgroups, most of it for the launch after next to become eligible —
 * 11.7 us per launch; three streams without any ordering (not a product) 11.15-11.3; three streams with a cross-stream wait
 * per two launches 11.9 — a barrier packet in a queue costs more than the third stream gains; the host waiting for the round
 * two back 11.5-11.8 (the queues run dry while it enqueues), four back 11.3; a fourth stream nothing more. */
#define AACG_OV_BUFFERS 16
#define AACG_PIPE_STREAMS 3

/* One workgroup's work: consecutive frames of one element of one stream.  The first run of a
 * chain holds up to 16 units (wave w = unit w, wave 0 starts from the overlap state); a later
 * run holds up to 15 units in waves 1..15 and wave 0 recomputes the tail of pred_unit.
 * The record is laid out for how a wave reads it — ONE scalar round trip: the header as eight dwords, and three dwords picked by
 * the wave's number (its unit, and where that unit's spectra and band words lie).  Between a wave's first instruction and its
 * first spectrum stand dependent memory round trips and nothing else, each 0.15 us on an idle memory system and 0.45 us under
 * streaming load (profiles/r05_budget.txt): with the addresses here the spectra are requested two round trips after the
 * start (kernel arguments, this record) instead of five (arguments, n_units, unit index, unit record, spectra), and the
 * unit record travels beside them. */
struct aacg_run {
    int32_t pred_unit;                /* -1: first run of its chain */
    int32_t n_units;
    int32_t is_last;                  /* last run of its chain: the final tail goes to the out buffer */
    uint32_t wave_nch;                /* 2 bits per WAVE: channels of the unit the wave loads first */
    int32_t ov0[2];                   /* per channel: float offset of its buffer 0 in the overlap pool (buffer r at + 1024 r) */
    int32_t rot[2];                   /* per channel: the buffer that held its state when the plan was made; launch number j of the
                                         plan (aacg_kparams.flip = j mod AACG_OV_BUFFERS) reads buffer (rot + flip) mod AACG_OV_BUFFERS and writes the next one */
    int32_t unit[AACG_RUN_W];
    /* per WAVE, not per unit (a predecessor wave shifts them, aacg_run_wave_unit): the unit the wave loads first — its index
     * (unit 0 for a wave without work: every wave loads, quant_load), its coefficient block and its band-word block
     * (aacg_unit_desc.coef_offset / meta_offset: copies, kept equal by everything that rewrites unit records) */
    int32_t  wave_unit[AACG_RUN_W];
    uint32_t wave_coef[AACG_RUN_W];
    uint32_t wave_meta[AACG_RUN_W];
};
static_assert(sizeof(aacg_run) == 288 && __builtin_offsetof(aacg_run, ov0) == 16 && __builtin_offsetof(aacg_run, wave_unit) == 96,
              "aacg_run: the header is read as eight dwords, the per-wave words by offset (imdct_run_body)");
/* the unit wave w of a run loads first: the predecessor in wave 0 of a later run (a full later run's wave 0 then takes its
 * own frame in a second pass), else its frame; -1: no work */
static inline int32_t aacg_run_wave_unit(const aacg_run& r, int w)
{
    if (r.pred_unit >= 0 && w == 0) return r.pred_unit;
    const int k = (r.pred_unit >= 0 && r.n_units < AACG_RUN_W) ? w - 1 : w;
    return k < r.n_units ? r.unit[k] : -1;
}

/* AACG_CCE_SPEC: one (coupling element, target channel) pair of one frame.  Dependent coupling (spectral domain,
 * cce.js:130-158): dst / src are coefficient blocks of the f32 spectrum buffer; independent (cce.js:121-128 with the
 * CCE's own filterbank output): src is the CCE's block in the side PCM buffer, dst the float offset of the target
 * channel's first sample in the PCM buffer, stride its n_out_ch. */
struct aacg_couple_job {
    uint32_t cce_unit;                /* the CCE's unit (window info, band types) */
    uint32_t src, dst, stride;
    uint32_t gain_off;                /* float index of the gain list in the plan's gain array */
    uint32_t reserved[3];
};

struct aacg_couple_params {
    const aacg_couple_job* jobs;
    const aacg_dev_unit*   units;
    const aacg_band_meta*  meta;      /* band types of the coupling elements (quantised input), or null */
    const aacg_tables*     tab;       /* the coefficient -> band maps (global memory) */
    const float*           gains;
    float*                 spec;      /* dependent coupling: the f32 spectra, in place */
    const float*           side;      /* independent coupling: the coupling elements' filterbank output, [block][1024] */
    float*                 pcm;       /* ... added into the interleaved PCM */
    int32_t                n_jobs;
    int32_t                reserved;
};

struct aacg_kparams {
    const aacg_dev_unit*  units;
    const aacg_run*       runs;
    const void*           coeffs;     /* float or int16_t, per input kind */
    const aacg_band_meta* meta;
    const aacg_dev_tns*   tns;        /* AACG_TNS_SPEC: indexed like aacg_unit_desc.tns_offset + c; else null */
    float*                pcm;
    float*                overlap;    /* overlap pool */
    float*                spec_out;   /* spectral-only kernel */
    const aacg_tables*    tab;
    int32_t               flip;       /* launches of this plan so far, mod AACG_OV_BUFFERS (aacg_run.rot) */
    int32_t               n_runs;
    int32_t               ablate;     /* -DAACG_PROFILE builds only (AACG_ABL in aacg_kernels.h); 0 otherwise */
    int32_t               reserved;
    float*                scratch;    /* [n_runs][2048]: parked predecessor tails of double-duty runs (last: the plain kernels never load it) */
    const aacg_pns_tables* pns;       /* AACG_PNS_SPEC: the spectral stage's noise tables */
};
/* AACG_CCE_SPEC kernels with the independent coupling in their epilogue (aacg_imdct_run_*_cpl) take three more pointers.  They
 * ride in fields those kernels have no other use for — no double duty (scratch), no optional stages (pns), no spectral
 * output (spec_out) — so that the kernel arguments of every other launch stay as they are (a longer argument block costs the
 * headline kernel its measured 0.05 us, and the layout of that block has cost it 0.35 us before: DESIGN.md 6). */
/* AACG_TNS_SPEC launches (the run kernels with the stages inside, the staged spectral kernel) take the plan's transition
 * matrices (tns_matrix_row, aacg_kernels.h) the same way: in `scratch`, which those launches have no other use for. */
static inline void aacg_set_tns_m(aacg_kparams* P, const double* m) { P->scratch = (float*)(void*)m; }
static inline void aacg_set_cpl(aacg_kparams* P, const aacg_couple_job* jobs, const float* gains, const float* side)
{
    P->scratch = (float*)(void*)jobs; P->pns = (const aacg_pns_tables*)(const void*)gains; P->spec_out = (float*)side;
}
#define AACG_CPL_JOBS(P)  ((const aacg_couple_job*)(const void*)(P).scratch)
#define AACG_CPL_GAINS(P) ((const float*)(const void*)(P).pns)
#define AACG_CPL_SIDE(P)  ((const float*)(P).spec_out)   /* the coupling elements' filterbank output, [block][1024], PCM-scaled */

/* ---- rendezvous cells: the runs of a chain in different workgroups (_rv builds) — and, for pipelined launches, in different
 * LAUNCHES ---- */
/* Every run of such a plan holds up to 16 frames and nobody recomputes anything: a later run's first frame and the run
 * before it meet in a rendezvous cell in global memory — whichever side arrives first publishes what it has (the windowed
 * tail, or the windowed first half) and leaves, the second finishes the frame.  Nobody waits for another workgroup, so no
 * dispatch order is assumed.  One link record per block. */
struct aacg_rv_link {
    int32_t link_in;                  /* cell through which the run before this one hands over; -1: first run of its chain */
    int32_t link_out;                 /* cell towards the next run; -1: last run of its chain */
    int32_t succ_unit;                /* first unit of the next run, or -1 */
    int32_t reserved;
};
#define AACG_RV_STATE_WORDS 2         /* per in-launch cell: the state word and a spare (16-byte records) */
#define AACG_RV_DATA_FLOATS 4096      /* per in-launch cell: [tail | head][channel][1024] */
#define AACG_RV_TAIL 1ull             /* state word = (epoch << 2) | one of these; any other value: nobody has been here in this epoch */
#define AACG_RV_HEAD 2ull

/* The same meeting between the LAST frame of a chain in one launch and its FIRST frame in the plan's next launch, when the
 * two launches overlap (aacg_decode_pipelined): one cell per (stream, channel, overlap buffer), used through the element's
 * first channel.  The tail payload of such a cell IS the overlap buffer (the state lands where a launch that comes later
 * and alone reads it plainly); the head payload has a pool of the same shape (aacg_rv_args.xl_head).  A consumer that leaves
 * its windowed first half also says where the finished samples go: its launch's PCM buffer is unknown to the launch before. */
struct aacg_xl_cell {
    unsigned long long state;         /* (epoch of the launch that WRITES this buffer << 2) | AACG_RV_TAIL / AACG_RV_HEAD */
    unsigned long long pcm;           /* with HEAD: address of sample 0 of the element's first channel in the consumer's frame */
    uint32_t n_out_ch;                /* with HEAD: that frame's interleave stride */
    uint32_t reserved[3];
};
struct aacg_rv_args {
    const aacg_rv_link*  links;       /* [n_runs], block order */
    unsigned long long*  state;       /* [n_links][AACG_RV_STATE_WORDS], epoch-tagged, never reset */
    float*               data;        /* [n_links][AACG_RV_DATA_FLOATS] */
    unsigned long long   epoch;       /* of this launch, never 0: tags its in-launch cells and the cross-launch cells it writes */
    aacg_xl_cell*        xl_cells;    /* [stream][channel][AACG_OV_BUFFERS]; null: chain ends read and write the overlap pool plainly */
    float*               xl_head;     /* [stream][channel][AACG_OV_BUFFERS][1024] */
    unsigned long long   epoch_in;    /* epoch of the launch that writes this launch's input state and may still be running;
                                         0: that state is complete (the launch is ordered behind its writer) */
};

/* ---- device front end (aacg_parse.h) ------------------------------------------------------------ */
#define AACG_PARSE_WG_SMALL   256      /* frames staged in LDS: shortest time per frame */
#define AACG_PARSE_WG_LARGE   1024     /* frames read in place, 16 waves per CU: highest rate on large batches */
#define AACG_PARSE_L1_BITS    9
#define AACG_PARSE_LUT_WORDS  12288     /* 12 first-level tables of 512 + the second-level tables */
#define AACG_PARSE_BUCKETS    1024      /* frame lengths in 8-byte steps, for the lane order */
#define AACG_PARSE_PAD_BYTES  32        /* readable bytes required after the last frame (look-ahead + 16-byte staging) */
/* LDS of a workgroup: tables, band columns, allocator word, then the arena */
#define AACG_PARSE_LDS_FIXED(lut_words, threads) ((size_t)(lut_words) * 4u + 160u + (size_t)AACG_MAX_SECTIONS * (threads) + 16u)
/* lut entry: bits 0..4 code length; bit 5 clear: bits 8..31 payload (scalefactor book: the value; spectral books:
 * values as 6-bit two's-complement fields);  bit 5 set: bits 0..4 = extra bits, bits 8..31 = index of a
 * second-level table */
typedef struct aacg_parse_tables {
    uint32_t lut[AACG_PARSE_LUT_WORDS];
    uint32_t lut_words;
    uint16_t swb_long[64];
    uint16_t swb_short[16];
    uint32_t n_swb_long, n_swb_short;
    float    tns_coef[4][16];           /* [2 * coef_compress + coef_res][field] (tns.js:50-63) */
} aacg_parse_tables;

typedef struct aacg_parse_params {
    const uint32_t* bytes;
    const aacg_parse_frame* frames;
    const aacg_parse_tables* tab;
    aacg_unit_desc* units;
    int16_t* q;
    aacg_band_meta* meta;
    aacg_tns_info* tns;
    aacg_parse_result* results;
    uint32_t n_frames, max_units, max_channels, options;
    uint32_t arena_bytes;      /* LDS left for staging the frames' bytes */
    uint32_t wg_threads;
    const uint32_t* order;     /* lane i parses frame order[i] (frames of similar length share a wave), or NULL: frame i */
} aacg_parse_params;

#endif
