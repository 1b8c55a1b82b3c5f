/*
 * aacg_parse_host.cpp — host side of the device front end: builds the lookup tables aacg_parse.h reads from the
 * code word lists the caller supplies (include/aacgpu.h, aacg_parser_create).  No HIP here; tests/emu links it too.
 */
#include "aacg_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

namespace {

const uint32_t kAlphabet[12] = {121, 81, 81, 81, 81, 81, 81, 64, 64, 169, 169, 289};   /* ISO/IEC 14496-3 4.A.1 */

uint32_t payload_of(int book, const aacg_code_entry& e)
{
    if (book == 0) return (uint32_t)(uint8_t)e.v[0];
    uint32_t p = 0;
    for (int j = 0; j < (book < 5 ? 4 : 2); j++) p |= ((uint32_t)e.v[j] & 63u) << (6 * j);
    return p;
}

}  // namespace

int aacg_parse_build_tables(int sample_index, const aacg_code_entry* entries, const uint32_t counts[12],
                            aacg_parse_tables* out, std::string* err)
{
    auto fail = [&](int rc, const std::string& m) { if (err) *err = m; return rc; };
    if (!entries || !counts || !out) return fail(AACG_ERR_INVALID_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    int off[64];
    int n = aacg_swb_offsets(sample_index, 1, off);
    if (n <= 0) return fail(AACG_ERR_INVALID_ARG, "sample_index out of range");
    out->n_swb_long = (uint32_t)n;
    for (int i = 0; i <= n; i++) out->swb_long[i] = (uint16_t)off[i];
    n = aacg_swb_offsets(sample_index, 0, off);
    out->n_swb_short = (uint32_t)n;
    for (int i = 0; i <= n; i++) out->swb_short[i] = (uint16_t)off[i];
    /* tns.js:50-63: -sin(q / iqfac) on a 3- or 4-bit grid, the field being q's low bits under coef_compress */
    for (int compress = 0; compress < 2; compress++)
        for (int res = 0; res < 2; res++) {
            const int bits = res + 3, fields = 1 << (bits - compress), half = 1 << (bits - 1);
            for (int i = 0; i < fields; i++) {
                const int s = i >= fields / 2 ? i - fields : i;
                out->tns_coef[2 * compress + res][i] = (float)-std::sin(s / ((s >= 0 ? half - 0.5 : half + 0.5) / (3.14159265358979323846 / 2.0)));
            }
        }

    const int L1 = AACG_PARSE_L1_BITS;
    uint32_t next = 12u << L1;
    const aacg_code_entry* e = entries;
    for (int book = 0; book < 12; e += counts[book], book++) {
        const std::string name = "codebook " + std::to_string(book);
        if (counts[book] != kAlphabet[book]) return fail(AACG_ERR_INVALID_ARG, name + ": " + std::to_string(counts[book]) + " entries, expected " + std::to_string(kAlphabet[book]));
        long double kraft = 0;
        std::map<uint32_t, std::vector<uint32_t>> deep;
        uint32_t* l1 = out->lut + ((size_t)book << L1);
        for (uint32_t s = 0; s < counts[book]; s++) {
            const int len = e[s].len;
            if (len < 1 || len > 24 || (e[s].code >> len) != 0) return fail(AACG_ERR_INVALID_ARG, name + ": malformed entry");
            kraft += std::ldexp(1.0L, -len);
            if (len <= L1) {
                const uint32_t lo = e[s].code << (L1 - len);
                for (uint32_t i = 0; i < (1u << (L1 - len)); i++) {
                    if (l1[lo + i]) return fail(AACG_ERR_INVALID_ARG, name + ": not a prefix code");
                    l1[lo + i] = (payload_of(book, e[s]) << 8) | (uint32_t)len;
                }
            } else {
                deep[e[s].code >> (len - L1)].push_back(s);
            }
        }
        if (kraft != 1.0L) return fail(AACG_ERR_INVALID_ARG, name + ": not a complete prefix code");
        for (auto& d : deep) {
            int extra = 0;
            for (uint32_t s : d.second) extra = std::max(extra, (int)e[s].len - L1);
            if (l1[d.first]) return fail(AACG_ERR_INVALID_ARG, name + ": not a prefix code");
            if (next + (1u << extra) > AACG_PARSE_LUT_WORDS) return fail(AACG_ERR_CAPACITY, name + ": lookup tables exceed AACG_PARSE_LUT_WORDS");
            for (uint32_t s : d.second) {
                const int len = e[s].len, tail_bits = len - L1;
                const uint32_t tail = e[s].code & ((1u << tail_bits) - 1u), lo = tail << (extra - tail_bits);
                for (uint32_t i = 0; i < (1u << (extra - tail_bits)); i++) {
                    if (out->lut[next + lo + i]) return fail(AACG_ERR_INVALID_ARG, name + ": not a prefix code");
                    out->lut[next + lo + i] = (payload_of(book, e[s]) << 8) | (uint32_t)len;
                }
            }
            l1[d.first] = (next << 8) | 0x20u | (uint32_t)extra;
            next += 1u << extra;
        }
    }
    out->lut_words = (next + 3u) & ~3u;            /* keeps what follows in LDS 16-byte aligned */
    return AACG_OK;
}
