/*
 * aacg_host.h — host-side internals shared by the engine and (for CPU-only tests of the
 * host logic) the lane emulator: table construction and the batch planner.  No HIP here.
 */
#ifndef AACG_HOST_H
#define AACG_HOST_H

#include <stdint.h>
#include <string>
#include <vector>

#include "aacg_device.h"

struct aacg_host_windows {
    float sine_long[1024], kbd_long[1024], sine_short[128], kbd_short[128];
};

/* Fills *t for config.sampleIndex (decoder.js:63); hw (optional) receives the plain windows. */
int aacg_build_tables(int sample_index, aacg_tables* t, aacg_host_windows* hw);
int aacg_build_pns_tables(int sample_index, aacg_pns_tables* t);
/* SWB_OFFSET_1024/128[sample_index] (tables.js:34-155): writes count+1 offsets, returns count. */
int aacg_swb_offsets(int sample_index, int is_long, int* dst);

/* One (stream, element) sequence of consecutive frames inside a batch. */
struct aacg_chain {
    uint32_t stream;
    uint16_t channel;
    uint8_t  n_ch;
    uint8_t  parity[2];     /* overlap buffer (0..AACG_OV_BUFFERS-1) holding the chain's input state when the plan was built */
    uint32_t first_run, n_runs;
};

struct aacg_plan_host {
    std::vector<aacg_dev_unit> units;   /* device copy of the units */
    std::vector<aacg_dev_tns>  tns;     /* device form of the TNS side info (AACG_TNS_SPEC), same indexing as the input */
    bool     any_tns = false;         /* some channel has AACG_CHAN_TNS_PRESENT and TNS records were given */
    bool     any_pns = false;         /* some unit carries AACG_UNIT_HAS_PNS */
    uint32_t short_units = 0;         /* units with an EIGHT_SHORT_SEQUENCE channel, as planned (or last refreshed from the host) */
    bool     needs_scratch = false;   /* some later run holds 16 frames: its first wave parks the predecessor's tails */
    bool     wide_frames = false;     /* at least half of the units belong to frames of more than two channels: the multichannel kernel variants (aacg_engine_nt.hip) */
    std::vector<aacg_run>   runs;     /* in launch (block) order, XCD-aware */
    /* the same chains cut for the rendezvous kernels (_rv builds): every run up to 16 frames, consecutive runs of a chain joined
     * by a rendezvous cell instead of a recomputed frame.  The route of plain batches with a chain longer than a run
     * (long_chains) and of every plain batch launched through aacg_decode_pipelined */
    bool     long_chains = false;
    bool     runs_moved = false;        /* aacg_plan_refresh_host: the batch's block offsets differ from the ones the run tables carried — they were rewritten */
    std::vector<aacg_run>     runs_rv;
    std::vector<aacg_rv_link> links_rv;     /* one per run, same order */
    uint32_t n_links_rv = 0;
    /* AACG_CCE_SPEC: independently switched coupling elements run through the filterbank like any channel, but into a
     * side buffer (cce_runs: their own launch); coupling jobs by coupling point and by round (round r: the r-th coupling
     * element of its frame, so that no two jobs of a round add to the same channel) */
    bool     any_cce = false;
    bool     any_cce_dependent = false;   /* some element couples in the spectral domain: the batch takes the staged f32 route */
    std::vector<aacg_run> cce_runs;
    uint32_t side_blocks = 0;         /* 1024-float blocks of the side PCM buffer */
    std::vector<aacg_couple_job> couple_jobs;                   /* sorted by (point, round) */
    std::vector<uint32_t> couple_first;                         /* couple_first[point][round] ... start indices, see aacg_plan.cpp */
    uint32_t couple_rounds = 0;
    std::vector<float> gains;                                   /* [n_cce][16][120] */
    /* independent coupling fused into the targets' epilogues (plans without double-duty runs): per-unit job lists
     * (aacg_dev_unit.cpl_first / cpl_n index fused_jobs); the AFTER_IMDCT entries of couple_first are then empty */
    bool     fused_independent = false;
    std::vector<aacg_couple_job> fused_jobs;
    uint32_t fused_first = 0;         /* where they start in couple_jobs (they are appended to it for the upload) */
    std::vector<aacg_chain> chains;
    bool     zero_fill = false;       /* some frame has a channel no unit writes (decoder.js:229-231) */
    uint32_t coef_blocks = 0;         /* 1 + highest (coef_offset + c) referenced */
    uint32_t meta_blocks = 0;
    size_t   pcm_floats = 0;          /* 1 + highest PCM float written */
};

/* float offset of overlap buffer `parity` (0..AACG_OV_BUFFERS-1) of (stream, channel) in the pool */
static inline int32_t aacg_ov_offset(int max_channels, uint32_t stream, uint32_t channel, int parity)
{
    return (int32_t)((((size_t)stream * (size_t)max_channels + channel) * (size_t)AACG_OV_BUFFERS + (unsigned)parity) * 1024u);
}

/* Validates the units and cuts them into runs.  parity: [max_streams * max_channels] current
 * input buffer (0..AACG_OV_BUFFERS-1) per (stream, channel), or NULL for all zero.  Returns AACG_OK or an error code
 * with a message in *err. */
int aacg_plan_build(const aacg_unit_desc* units, uint32_t n_units, int sample_index,
                    int max_streams, int max_channels, const uint8_t* parity,
                    aacg_plan_host* out, std::string* err,
                    const aacg_tns_info* tns = nullptr, uint32_t n_tns = 0,
                    const aacg_cce_info* cce = nullptr, uint32_t n_cce = 0);

/* A kept plan's unit records rewritten from the next batch's (same structure: streams, frames, elements, PCM positions);
 * AACG_ERR_LAYOUT_CHANGE if the structure differs, in which case nothing has been touched. */
void aacg_plan_fill_run_waves(aacg_plan_host* h);
int aacg_plan_refresh_host(aacg_plan_host* h, const aacg_unit_desc* units, uint32_t n_units, int sample_index, bool tns_spec, std::string* err);

/* tns.js:111-152: per-filter sample range and LPC coefficients of one channel (float32 stores as in the
 * reference's Float32Array lpc).  Returns AACG_OK or AACG_ERR_UNSUPPORTED (order > 12). */
int aacg_tns_prepare(int sample_index, const aacg_chan_info* info, const aacg_tns_info* in, aacg_dev_tns* out);

/* Device front end: lookup tables from the caller's (length, code word, values) lists; every book is checked to be
 * a complete prefix code of the standard's alphabet size.  Returns AACG_OK or AACG_ERR_INVALID_ARG / _CAPACITY. */
int aacg_parse_build_tables(int sample_index, const aacg_code_entry* entries, const uint32_t counts[12],
                            aacg_parse_tables* out, std::string* err);

#endif
