/*
 * aacg_codebooks.cpp — the 12 AAC Huffman codebooks (ISO/IEC 14496-3 tables 4.A.1-4.A.12) as aacg_code_entry
 * records: aacg_standard_codebooks() of include/aacgpu.h.  The reference keeps the same facts of the standard as
 * the private arrays of src/huffman.js:22-1418; here they are stored in index order as (length, code word) only
 * (aacg_codebook_data.inc, written by tools/gen/gen_codebooks.js) and the values follow from the index.
 */
#include "aacg_host.h"

#include <cstring>

namespace {
#include "aacg_codebook_data.inc"
}

extern "C" uint32_t aacg_standard_codebooks(aacg_code_entry* entries, uint32_t counts[12])
{
    uint32_t total = 0;
    const uint32_t* w = kBookWords;
    for (int book = 0; book < 12; book++) {
        const int dim = kBookShape[book][0], is_signed = kBookShape[book][1], mod = kBookShape[book][2];
        uint32_t n = 1;
        for (int j = 0; j < dim; j++) n *= (uint32_t)mod;
        if (counts) counts[book] = n;
        if (entries)
            for (uint32_t idx = 0; idx < n; idx++) {
                aacg_code_entry& e = entries[total + idx];
                std::memset(&e, 0, sizeof e);
                e.code = w[idx] & 0xFFFFFFu;
                e.len = (uint8_t)(w[idx] >> 24);
                uint32_t r = idx;
                for (int j = dim - 1; j >= 0; j--, r /= (uint32_t)mod)          /* first value most significant */
                    e.v[j] = (int8_t)((int)(r % (uint32_t)mod) - (is_signed ? (mod - 1) / 2 : 0));
            }
        w += n;
        total += n;
    }
    return total;
}
