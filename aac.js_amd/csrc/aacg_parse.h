/*
 * aacg_parse.h — the raw_data_block parser as device code: one lane parses one frame.
 *
 * What aac.js does serially per frame between `stream.peek(12)` and `this.process(elements)` (decoder.js:126-201:
 * element loop; ics.js:56-201,279-314; cpe.js:37-75; tns.js:68-103; cce.js:45-119; huffman.js:1425-1490) is
 * independent from frame to frame, so a batch of frames is parsed by as many lanes.  Each lane walks its own bit
 * string — the control flow diverges inside a wave, which is what makes this stage slow per lane and the reason
 * to give it thousands of lanes — and writes the engine's input directly: aacg_unit_desc records, int16 spectra,
 * band words, TNS records (include/aacgpu.h).  Nothing is dequantised here.
 *
 * Code words: a two-level lookup table per codebook (9-bit first level), all 12 in one array staged to LDS; an
 * entry carries the code length and the decoded values, so one LDS read resolves a code word of <= 9 bits.
 * The host builds the array from (length, code word, values) lists it is given (aacg_parse_build_tables,
 * aacg_parse_host.cpp); the lists themselves are not part of this repository (aac.js_amd/js/codebooks.js).
 *
 * Written against devport.h like the transform kernels, and executed lane by lane on the CPU by tests/emu.
 * Errors do not unwind: the lane's status is set, every later read returns 0, every loop checks the status.
 */
#ifndef AACG_PARSE_H
#define AACG_PARSE_H

#include "aacg_device.h"

#if defined(AACG_EMU_BUILD)
#include "devport_emu.h"
#else
#include "devport.h"
#endif

namespace aacg_parse {

struct bit_reader {
    int lds;                   /* LDS byte address of the frame's staged bytes (big-endian dwords), or -1: */
    const uint32_t* glb;       /* ... read in place.  Kept apart so that the staged case compiles to ds_read: a flat or
                                  global load would wait on vmcnt, i.e. for every spectrum store issued before it */
    uint32_t pos, end;         /* absolute bit positions */
    uint32_t cached;           /* dword index held in w */
    uint64_t w;
    int status;
    int deferred;              /* a refusal that only counts if the rest of the frame parses: pulse data without
                                  AACG_PARSE_APPLY_PULSES, a TNS order the records cannot hold */
};

DP_DEVICE uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

DP_DEVICE void br_open(bit_reader& r, int lds, const uint32_t* glb, uint32_t byte_offset, uint32_t byte_length)
{
    r.lds = lds; r.glb = glb; r.pos = byte_offset * 8u; r.end = r.pos + byte_length * 8u;
    r.cached = 0xffffffffu; r.w = 0; r.status = AACG_PARSE_OK; r.deferred = 0;
}

/* n = 1..32 bits at the current position (zeros past the end of the buffer's padding are the host's job) */
DP_DEVICE uint32_t br_peek(bit_reader& r, int n)
{
    const uint32_t i = r.pos >> 5;
    if (i != r.cached) {
        if (r.lds >= 0) r.w = ((uint64_t)be32(dp_lds_read_u32(r.lds + 4 * (int)i)) << 32) | be32(dp_lds_read_u32(r.lds + 4 * (int)i + 4));
        else       r.w = ((uint64_t)be32(r.glb[i]) << 32) | be32(r.glb[i + 1]);
        r.cached = i;
    }
    return (uint32_t)((r.w << (r.pos & 31u)) >> (64 - n));
}

DP_DEVICE void br_skip(bit_reader& r, uint32_t n)
{
    if (r.status) return;
    if (n > r.end - r.pos) { r.status = AACG_PARSE_INSUFFICIENT_DATA; return; }
    r.pos += n;
}

DP_DEVICE uint32_t br_read(bit_reader& r, int n)
{
    if (r.status) return 0;
    if ((uint32_t)n > r.end - r.pos) { r.status = AACG_PARSE_INSUFFICIENT_DATA; return 0; }
    const uint32_t v = br_peek(r, n);
    r.pos += (uint32_t)n;
    return v;
}

DP_DEVICE void br_fail(bit_reader& r, int code) { if (!r.status) r.status = code; }

/* the next 32 bits, for a code word and its sign bits in one look */
DP_DEVICE uint32_t br_peek32(bit_reader& r)
{
    const uint32_t i = r.pos >> 5;
    if (i != r.cached) {
        if (r.lds >= 0) r.w = ((uint64_t)be32(dp_lds_read_u32(r.lds + 4 * (int)i)) << 32) | be32(dp_lds_read_u32(r.lds + 4 * (int)i + 4));
        else       r.w = ((uint64_t)be32(r.glb[i]) << 32) | be32(r.glb[i + 1]);
        r.cached = i;
    }
    return (uint32_t)((r.w << (r.pos & 31u)) >> 32);
}

/* entry of the code word at the top of `win` */
DP_DEVICE uint32_t lut_entry(int lut, int book, uint32_t win)
{
    uint32_t e = dp_lds_read_u32(lut + 4 * (book * (1 << AACG_PARSE_L1_BITS) + (int)(win >> (32 - AACG_PARSE_L1_BITS))));
    if (e & 0x20u) {
        const int extra = (int)(e & 31u);
        e = dp_lds_read_u32(lut + 4 * (int)((e >> 8) + ((win >> (32 - AACG_PARSE_L1_BITS - extra)) & ((1u << extra) - 1u))));
    }
    return e;
}

/* next code word of `book`: returns the entry's 24-bit payload */
DP_DEVICE uint32_t huff(bit_reader& r, int lut, int book)
{
    if (r.status) return 0;
    const uint32_t e = lut_entry(lut, book, br_peek32(r));
    br_skip(r, e & 31u);
    return r.status ? 0u : e >> 8;
}

DP_DEVICE int field6(uint32_t payload, int j) { return (int)(payload << (26 - 6 * j)) >> 26; }

struct ics_info {
    int seq, shape, max_sfb, groups, n_swb;
    uint32_t group_len;        /* 8 x 4 bits */
};
DP_DEVICE int group_len(const ics_info& i, int g) { return (int)((i.group_len >> (4 * g)) & 15u); }

struct lane_ctx {
    const aacg_parse_params* P;
    /* LDS byte addresses (explicit ds_read / ds_write through devport, never a flat access) */
    int lut;
    int swb_long, swb_short;   /* uint16 offsets */
    int bands;                 /* this lane's byte column: byte idx * stride = band type | ms_used << 7 */
    int stride;
    unsigned char* arena;      /* the workgroup's frames are staged here as far as they fit */
    int* arena_top;
    int arena_bytes;
};

/* ics_info (ics.js:279-314) */
DP_DEVICE void parse_ics_info(bit_reader& r, const lane_ctx& c, ics_info& info)
{
    br_skip(r, 1);
    info.seq = (int)br_read(r, 2);
    info.shape = (int)br_read(r, 1);
    info.groups = 1; info.group_len = 1;
    if (info.seq == 2) {
        info.max_sfb = (int)br_read(r, 4);
        const uint32_t bits = br_read(r, 7);
        for (int i = 6; i >= 0; i--) {
            if ((bits >> i) & 1u) info.group_len += 1u << (4 * (info.groups - 1));
            else { info.group_len |= 1u << (4 * info.groups); info.groups++; }
        }
        info.n_swb = c.P->tab->n_swb_short;
    } else {
        info.max_sfb = (int)br_read(r, 6);
        if (br_read(r, 1)) br_fail(r, AACG_PARSE_PREDICTION);
        info.n_swb = c.P->tab->n_swb_long;
    }
    if (info.max_sfb > info.n_swb) br_fail(r, AACG_PARSE_MAX_SFB);
}

/* tns_data (tns.js:68-103) -> aacg_tns_info, or just consumed when out == nullptr */
DP_DEVICE void parse_tns(bit_reader& r, const lane_ctx& c, const ics_info& info, aacg_tns_info* out)
{
    const bool is_short = info.seq == 2;
    const int windows = is_short ? 8 : 1, n_bits = is_short ? 1 : 2, len_bits = is_short ? 4 : 6, ord_bits = is_short ? 3 : 5;
    if (out) for (int w = 0; w < 8; w++) out->n_filt[w] = 0;
    for (int w = 0; w < windows && !r.status; w++) {
        const int nf = (int)br_read(r, n_bits);
        if (out) out->n_filt[w] = (uint8_t)nf;
        if (!nf) continue;
        const int res = (int)br_read(r, 1);
        for (int f = 0; f < nf && !r.status; f++) {
            const int length = (int)br_read(r, len_bits), order = (int)br_read(r, ord_bits);
            if (order > 20) { br_fail(r, AACG_PARSE_TNS_ORDER); return; }
            if (out && order > AACG_TNS_MAX_ORDER && !r.deferred) r.deferred = AACG_PARSE_TNS_ORDER;
            aacg_tns_filter* flt = out && order <= AACG_TNS_MAX_ORDER ? &out->filt[is_short ? w : f] : nullptr;
            if (flt) { flt->length = (uint8_t)length; flt->order = (uint8_t)order; flt->direction = 0; flt->reserved = 0; }
            if (!order) continue;
            const int direction = (int)br_read(r, 1), compress = (int)br_read(r, 1), width = res + 3 - compress;
            if (flt) flt->direction = (uint8_t)direction;
            for (int i = 0; i < order; i++) {
                const uint32_t field = br_read(r, width);
                if (flt) flt->coef[i] = c.P->tab->tns_coef[2 * compress + res][field];
            }
        }
    }
}

struct ics_out {
    int16_t* q;                /* this channel's 1024 values (pre-zeroed), or nullptr: parse and drop */
    uint16_t* meta;            /* this channel's 120 band words (pre-zeroed: only the coded ones are written) */
    aacg_tns_info* tns;        /* or nullptr */
    aacg_chan_info* chan;
};

/* individual_channel_stream (ics.js:56-201).  have_info: common_window, `info` was read by the caller.
 * Returns flags: 1 noise bands, 2 TNS.  (info travels by reference to a plain local: a pointer chosen at run time
 * would put it in scratch memory, and every scratch access waits for the spectrum stores before it.) */
DP_DEVICE int parse_ics(bit_reader& r, const lane_ctx& c, bool have_info, ics_info& info, const ics_out& o, bool keep_ms)
{
    const int global_gain = (int)br_read(r, 8);
    if (!have_info) parse_ics_info(r, c, info);
    if (r.status) return 0;
    const int groups = info.groups, max_sfb = info.max_sfb, nb = groups * max_sfb, S = c.stride;
    const bool is_short = info.seq == 2;

    /* section_data (ics.js:83-116) */
    const int len_bits = is_short ? 3 : 5, esc = (1 << len_bits) - 1;
    for (int g = 0, idx = 0; g < groups && !r.status; g++)
        for (int k = 0; k < max_sfb && !r.status;) {
            const int bt = (int)br_read(r, 4);
            if (bt == 12) { br_fail(r, AACG_PARSE_BAND_TYPE); break; }
            int end = k, incr;
            while ((incr = (int)br_read(r, len_bits)) == esc && !r.status) end += incr;
            end += incr;
            if (end > max_sfb) { br_fail(r, AACG_PARSE_TOO_MANY_BANDS); break; }
            for (; k < end; k++, idx++) dp_lds_write_u8(c.bands + idx * S, (keep_ms ? dp_lds_read_u8(c.bands + idx * S) & 0x80u : 0u) | (uint32_t)bt);
        }
    if (r.status) return 0;

    /* scale_factor_data (ics.js:118-173) */
    int sf_spec = global_gain, sf_noise = global_gain - 90, sf_int = 0, flags = 0;
    bool first_noise = true;
    for (int idx = 0; idx < nb && !r.status; idx++) {
        const int b = (int)dp_lds_read_u8(c.bands + idx * S), bt = b & 15;
        int word = bt << 12;
        if (bt == 0) {
        } else if (bt >= 14) {
            sf_int += (int)huff(r, c.lut, 0) - 60;
            word |= 200 - (sf_int < -155 ? -155 : sf_int > 100 ? 100 : sf_int);
        } else if (bt == 13) {
            if (first_noise) { sf_noise += (int)br_read(r, 9) - 256; first_noise = false; }
            else sf_noise += (int)huff(r, c.lut, 0) - 60;
            word |= (200 + (sf_noise < -100 ? -100 : sf_noise > 155 ? 155 : sf_noise)) | AACG_META_NEGATE;
            flags |= 1;
        } else {
            sf_spec += (int)huff(r, c.lut, 0) - 60;
            if (sf_spec > 255 || sf_spec < -100) { br_fail(r, AACG_PARSE_SCALEFACTOR); break; }
            word |= sf_spec + 100;
        }
        if (b & 0x80) word |= AACG_META_MS_USED;
        if (o.meta) o.meta[idx] = (uint16_t)word;
    }

    /* pulse_data (ics.js:175-201): kept in registers until the spectrum is there */
    int n_pulse = 0;
    uint64_t pulse_at = 0;         /* 4 x 16 bits */
    uint32_t pulse_amp = 0;        /* 4 x 4 bits */
    if (br_read(r, 1)) {
        if (is_short) br_fail(r, AACG_PARSE_PULSE_IN_SHORT);
        n_pulse = (int)br_read(r, 2) + 1;
        const int swb = (int)br_read(r, 6);
        if (swb >= info.n_swb) br_fail(r, AACG_PARSE_PULSE_RANGE);
        uint32_t at = r.status ? 0u : dp_lds_read_u16(c.swb_long + 2 * swb);
        for (int i = 0; i < 4; i++)
            if (i < n_pulse) {
                at += br_read(r, 5);
                if (at > 1023u) br_fail(r, AACG_PARSE_PULSE_RANGE);
                pulse_at |= (uint64_t)(at & 1023u) << (16 * i);
                pulse_amp |= br_read(r, 4) << (4 * i);
            }
        if (!(c.P->options & AACG_PARSE_APPLY_PULSES) && !r.deferred) r.deferred = AACG_PARSE_PULSE_DATA;
    }

    if (br_read(r, 1)) { parse_tns(r, c, info, o.tns); flags |= 2; }
    if (br_read(r, 1)) br_fail(r, AACG_PARSE_GAIN_CONTROL);
    if (r.status) return 0;

    if (o.chan) {
        o.chan->window_sequence = (uint8_t)info.seq; o.chan->window_shape = (uint8_t)info.shape; o.chan->window_shape_prev = 0;
        o.chan->max_sfb = (uint8_t)max_sfb; o.chan->group_count = (uint8_t)groups; o.chan->flags = (flags & 2) ? AACG_CHAN_TNS_PRESENT : 0;
        o.chan->reserved[0] = o.chan->reserved[1] = 0;
        for (int g = 0; g < 8; g++) o.chan->group_len[g] = (uint8_t)(g < groups ? group_len(info, g) : 0);
    }

    /* spectral_data (ics.js:203-261 without the dequantisation).  ONE loop per lane, one code word or one step of
     * the (group, band, window) walk per trip: written as the reference's four nested loops, a wave would run every
     * level for as many trips as its slowest lane needs at that level — the product of the maxima instead of the
     * maximum of the sums.  Quads and pairs share the body (a pair's fields 2, 3 are zero and take no sign bits). */
    if (nb > 0) {
        const int off = is_short ? c.swb_short : c.swb_long;
        int g = 0, sfb = -1, idx = -1, glen = group_len(info, 0), w = glen, group_off = 0;
        int k = 0, hi = 0, bt = 0, lo0 = 0, hi0 = 0;
        uint32_t sgn = 0, held = 0;
        while (!r.status) {
            if (k >= hi) {
                bool done = false;
                for (;;) {                                                  /* to the next coded (band, window) */
                    if (++w < glen) break;
                    w = 0; sfb++; idx++;
                    if (sfb >= max_sfb) {
                        sfb = 0; group_off += glen * 128;
                        if (++g >= groups) { done = true; break; }
                        glen = group_len(info, g);
                    }
                    bt = (int)dp_lds_read_u8(c.bands + idx * S) & 15;
                    if (bt == 0 || bt >= 13) { w = glen; continue; }           /* nothing coded here */
                    lo0 = (int)dp_lds_read_u16(off + 2 * sfb); hi0 = (int)dp_lds_read_u16(off + 2 * sfb + 2);
                    sgn = (bt >= 7 || bt == 3 || bt == 4) ? 1u : 0u;
                    break;
                }
                if (done) break;
                k = group_off + w * 128 + lo0; hi = group_off + w * 128 + hi0;      /* bands are >= 4 wide: k < hi, decode in this trip */
            }
            const uint32_t win = br_peek32(r), e = lut_entry(c.lut, bt, win), p = e >> 8;
            uint32_t len = e & 31u, sb = win << len;                         /* sign bits follow the code word: <= 19 + 4 bits */
            int v0 = field6(p, 0), v1 = field6(p, 1), v2 = field6(p, 2), v3 = field6(p, 3);
            {   /* one sign bit per non-zero value of an unsigned book, in order; branch-free: the lanes of a wave
                 * hold different values, every `if` here would run both ways */
                const uint32_t n0 = sgn & (uint32_t)(v0 != 0), n1 = sgn & (uint32_t)(v1 != 0), n2 = sgn & (uint32_t)(v2 != 0), n3 = sgn & (uint32_t)(v3 != 0);
                const int s0 = -(int)((sb >> 31) & n0); sb <<= n0;
                const int s1 = -(int)((sb >> 31) & n1); sb <<= n1;
                const int s2 = -(int)((sb >> 31) & n2); sb <<= n2;
                const int s3 = -(int)((sb >> 31) & n3);
                v0 = (v0 ^ s0) - s0; v1 = (v1 ^ s1) - s1; v2 = (v2 ^ s2) - s2; v3 = (v3 ^ s3) - s3;
                len += n0 + n1 + n2 + n3;
            }
            br_skip(r, len);
            if (bt == 11 && (((v0 + 16) & ~32) == 0 || ((v1 + 16) & ~32) == 0)) {      /* |v| == 16: escape sequences follow, v0's first */
                if (v0 == 16 || v0 == -16) {
                    int n = 4;
                    while (br_read(r, 1)) n++;
                    if (n > 12) br_fail(r, AACG_PARSE_ESCAPE);
                    const int mag = (1 << (n & 15)) + (int)br_read(r, n & 15);
                    v0 = v0 < 0 ? -mag : mag;
                }
                if (v1 == 16 || v1 == -16) {
                    int n = 4;
                    while (br_read(r, 1)) n++;
                    if (n > 12) br_fail(r, AACG_PARSE_ESCAPE);
                    const int mag = (1 << (n & 15)) + (int)br_read(r, n & 15);
                    v1 = v1 < 0 ? -mag : mag;
                }
            }
            /* always 8 bytes per store: a quad as it is, a pair together with its neighbour (runs start on multiples of 4
             * and are multiples of 4 wide, so pairs come in twos) — half the write requests, the ceiling of large batches */
            const uint32_t lo = (uint32_t)(v0 & 0xffff) | ((uint32_t)v1 << 16);
            if (o.q && !r.status) {
                if (bt < 5) { dpu2 d; d.x = lo; d.y = (uint32_t)(v2 & 0xffff) | ((uint32_t)v3 << 16); *(dpu2*)(o.q + k) = d; }
                else if (k & 2) { dpu2 d; d.x = held; d.y = lo; *(dpu2*)(o.q + k - 2) = d; }
            }
            held = lo;
            k += bt < 5 ? 4 : 2;
        }
    }
    if (r.status) return 0;
    if (o.q)
        for (int i = 0; i < 4; i++)
            if (i < n_pulse) {
                const int amp = (int)((pulse_amp >> (4 * i)) & 15u), at = (int)((pulse_at >> (16 * i)) & 1023u), v = o.q[at];
                o.q[at] = (int16_t)(v > 0 ? v + amp : v - amp);
            }
    return flags;
}

/* coupling_channel_element: consume the bits the reference consumes (cce.js:45-119), or the standard's */
DP_DEVICE void parse_cce(bit_reader& r, const lane_ctx& c)
{
    int point = 2 * (int)br_read(r, 1), gains = 0;
    const int coupled = (int)br_read(r, 3);
    for (int i = 0; i <= coupled && !r.status; i++) {
        gains++;
        const int pair = (int)br_read(r, 1);
        br_skip(r, 4);
        if (pair && br_read(r, 2) == 3) gains++;
    }
    point += (int)br_read(r, 1);
    point |= point >> 1;
    br_skip(r, 3);
    ics_info info;
    const ics_out none = {nullptr, nullptr, nullptr, nullptr};
    parse_ics(r, c, false, info, none, false);
    if (r.status) return;
    const bool quirks = (c.P->options & AACG_PARSE_REFERENCE_QUIRKS) != 0, after = !quirks && point == 3;
    const int nb = info.groups * info.max_sfb;
    for (int i = 0; i < gains && !r.status; i++) {
        int cge = 1;
        if (i > 0) {
            cge = after ? 1 : (int)br_read(r, 1);
            if (cge) huff(r, c.lut, 0);
        }
        if (after) continue;
        for (int b = 0, idx = 0; b < nb && !r.status; b++) {
            const bool coded = (dp_lds_read_u8(c.bands + (quirks ? idx : b) * c.stride) & 15u) != 0;
            if (coded && cge == 0) huff(r, c.lut, 0);
            if (coded) idx++;
        }
    }
}

/* one frame: raw_data_block, after an optional ADTS header (decoder.js:129-200) */
DP_DEVICE void parse_frame(const lane_ctx& c, uint32_t frame)
{
    const aacg_parse_params& P = *c.P;
    /* The frame's bytes go to LDS first (16-byte pieces, 8 bytes of look-ahead included), so that refilling the bit
     * window is a ds_read instead of a global load somewhere in the wave at nearly every code word (measured: about
     * 20 % off the time per launch).  A frame that does not fit any more is read in place. */
    const uint32_t off = P.frames[frame].byte_offset, len = P.frames[frame].byte_length;
    const uint32_t first = off & ~15u, span = ((off + len + 8u + 15u) & ~15u) - first;
    const unsigned char* src = (const unsigned char*)P.bytes + first;
    int staged = -1;
    const int slot = dp_lds_atomic_add(c.arena_top, (int)span);
    if (slot >= 0 && slot + (int)span <= c.arena_bytes) {
        dpi4* dst = (dpi4*)(c.arena + slot);
        for (uint32_t k = 0; k < span / 16u; k++) dst[k] = ((const dpi4*)src)[k];
        staged = dp_lds_addr(dst);
    }
    bit_reader r;
    br_open(r, staged, (const uint32_t*)src, off - first, len);
    const uint32_t start = r.pos;
    if (r.end - r.pos >= 56 && br_peek(r, 12) == 0xfffu) {           /* ADTS header (adts_demuxer.js:28-52) */
        br_skip(r, 15);
        const int protection_absent = (int)br_read(r, 1);
        br_skip(r, protection_absent ? 40 : 56);
    }
    int n_units = 0, channel = 0, any = 0;
    bool over = false;
    for (;;) {
        const int type = (int)br_read(r, 3);
        if (r.status || type == 7) break;
        int id = (int)br_read(r, 4);
        if (type == 0 || type == 3 || type == 1) {
            const int n_ch = type == 1 ? 2 : 1;
            /* beyond what the caller allowed for: the element is still parsed (its own errors come first), nothing is written */
            if (n_units >= (int)P.max_units || channel + n_ch > (int)P.max_channels) over = true;
            const uint32_t block = frame * P.max_channels + (uint32_t)channel;
            aacg_unit_desc* u = over ? nullptr : &P.units[frame * P.max_units + (uint32_t)n_units];
            ics_info info;
            int unit_flags = 0, fl = 0;
            bool ms = false;
            if (n_ch == 2 && br_read(r, 1)) {
                unit_flags |= AACG_UNIT_COMMON_WINDOW;
                parse_ics_info(r, c, info);
                const int mask = (int)br_read(r, 2);
                if (mask == 3) br_fail(r, AACG_PARSE_MS_MASK);
                if (mask && !r.status) {
                    unit_flags |= AACG_UNIT_MASK_PRESENT;
                    ms = true;
                    for (int i = 0; i < info.groups * info.max_sfb; i++) dp_lds_write_u8(c.bands + i * c.stride, (mask == 2 ? 1u : br_read(r, 1)) << 7);
                }
            }
            for (int k = 0; k < n_ch && !r.status; k++) {
                const ics_out none = {nullptr, nullptr, nullptr, nullptr};
                const ics_out o = over ? none : ics_out{ P.q + (size_t)(block + k) * 1024u, P.meta[block + k].band, P.tns ? &P.tns[block + k] : nullptr, &u->ch[k] };
                fl |= parse_ics(r, c, (unit_flags & AACG_UNIT_COMMON_WINDOW) != 0, info, o, ms && k == 0);
            }
            if (r.status) break;
            if (over) continue;
            if (n_ch == 1) { aacg_chan_info* z = &u->ch[1]; for (int i = 0; i < 16; i++) ((uint8_t*)z)[i] = 0; }
            u->stream = 0; u->pcm_offset = 0; u->channel = (uint16_t)channel; u->n_out_ch = 0;
            u->n_ch = (uint8_t)n_ch; u->flags = (uint8_t)(unit_flags | ((fl & 1) ? AACG_UNIT_HAS_PNS : 0)); u->reserved0 = (uint16_t)((type << 4) | id);
            u->coef_offset = block; u->meta_offset = block; u->tns_offset = (fl & 2) ? block : 0; u->reserved1 = 0;
            any |= fl;
            n_units++; channel += n_ch;
        } else if (type == 2) {
            parse_cce(r, c);
            any |= AACG_PARSE_HAS_CCE;             /* the element's bits were consumed and nothing of it kept (decoder.js:406-433 never applies it) */
        } else if (type == 4) {
            const int align = (int)br_read(r, 1);
            int count = (int)br_read(r, 8);
            if (count == 255) count += (int)br_read(r, 8);
            if (align) br_skip(r, (0u - r.pos) & 7u);
            br_skip(r, (uint32_t)count * 8u);
        } else if (type == 5) {
            br_fail(r, AACG_PARSE_PCE);
        } else {
            if (id == 15) id += (int)br_read(r, 8) - 1;
            br_skip(r, (uint32_t)id * 8u);
        }
    }
    if (!r.status) br_skip(r, (0u - r.pos) & 7u);
    if (r.deferred) br_fail(r, r.deferred);
    if (over) br_fail(r, AACG_PARSE_CAPACITY);
    aacg_parse_result* res = &P.results[frame];
    res->status = (uint8_t)r.status; res->n_units = (uint8_t)(r.status ? 0 : n_units); res->n_channels = (uint8_t)(r.status ? 0 : channel);
    res->flags = (uint8_t)any;
    res->bits_used = r.pos - start;
}

/* kernel body: workgroup of P.wg_threads lanes; lane t of block b parses frame order[b * threads + t].  A wave takes as
 * long as its longest frame, so the launcher sorts the frames by length and deals the sorted 64-frame pieces out to the
 * workgroups in turn: the lanes of a wave finish together, and every CU gets long and short waves alike. */
DP_DEVICE void parse_body(const aacg_parse_params& P)
{
    uint32_t* lds = (uint32_t*)dp_lds();
    const aacg_parse_tables* T = P.tab;
    const int tid = dp_tid(), words = (int)T->lut_words, threads = (int)P.wg_threads;
    for (int i = tid; i < words; i += threads) lds[i] = T->lut[i];
    uint16_t* swb = (uint16_t*)(lds + words);
    for (int i = tid; i < 64 + 16; i += threads) swb[i] = i < 64 ? T->swb_long[i] : T->swb_short[i - 64];
    unsigned char* bands = (unsigned char*)(swb + 80);
    int* top = (int*)(bands + AACG_MAX_SECTIONS * threads);
    unsigned char* arena = (unsigned char*)(top + 4);
    if (tid == 0) *top = 0;
    dp_block_sync();
    const uint32_t lane = (uint32_t)dp_block() * (uint32_t)threads + (uint32_t)tid;
    const uint32_t frame = P.order ? P.order[lane] : lane;       /* the order covers every lane of the grid; 0xffffffff = idle */
    if (frame >= P.n_frames) return;
    const lane_ctx c = { &P, dp_lds_addr(lds), dp_lds_addr(swb), dp_lds_addr(swb + 64), dp_lds_addr(bands + tid), threads, arena, top, (int)P.arena_bytes };
    parse_frame(c, frame);
}

}  // namespace aacg_parse

#endif
