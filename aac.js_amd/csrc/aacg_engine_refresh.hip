/*
 * aacg_engine_refresh.hip — a plan's unit records rewritten on the device from what aacg_parse_device produced
 * (aacg_plan_refresh_from_parse, include/aacgpu.h).  The run tables of a plan only depend on which streams bring
 * how many frames of which element layout; consecutive batches of the same streams share them.  What changes from
 * batch to batch is what the parser found per frame — window sequence and shape, max_sfb, grouping, flags — and that
 * never has to visit the host: one lane per unit copies it from the parser's record into the plan's device record,
 * keeps the planner's part (stream, PCM offset, channel layout), derives the group-of-window map, and turns a frame
 * the parser refused into a silent one (ONLY_LONG, max_sfb 0), counting it.
 *
 * Round 6: plan unit i need not be parsed record i.  With a map (aacg_refresh_map, include/aacgpu.h) a plan lists only the
 * elements its streams really have — streams of different element layouts in one batch (SCE + CPE + CPE + LFE beside plain
 * stereo; decoder.js:233-247 walks whatever elements a frame brings) — and says for each where the parser put it and how many
 * elements its frame must have; a frame with another count is not the frame the plan was made for and goes silent as a whole.
 */
#include <hip/hip_runtime.h>

#include "aacg_device.h"

static __device__ bool unit_matches(const aacg_unit_desc& have, const aacg_unit_desc& p, int refuse_pns)
{
    /* (the blocks: the plan's run tables carry copies of the offsets, aacg_run.wave_coef — a frame the parser put elsewhere is not
     * the frame the plan was made for) */
    return p.n_ch == have.n_ch && p.channel == have.channel && p.coef_offset == have.coef_offset && p.meta_offset == have.meta_offset &&
           !(refuse_pns && (p.flags & AACG_UNIT_HAS_PNS));              /* AACG_PNS_REFERENCE engines do not decode noise bands */
}

extern "C" __global__ __launch_bounds__(256)
void aacg_units_refresh(aacg_dev_unit* units, const aacg_unit_desc* parsed, aacg_parse_result* results, const aacg_refresh_map* map,
                        uint32_t n_units, uint32_t max_units, int refuse_pns, uint32_t* refused)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_units) return;
    aacg_dev_unit u = units[i];
    /* frame_units: elements the frame must have | elements of them the plan lists << 8 (0: all; aacgpu.h) */
    const uint32_t src = map ? map[i].parsed_index : i, want = map ? (map[i].frame_units & 0xffu) : 0u;
    const uint32_t listed = map && (map[i].frame_units >> 8) ? ((map[i].frame_units >> 8) & 0xffu) : want;
    const aacg_unit_desc p = parsed[src];
    const aacg_parse_result r = results[src / max_units];
    const uint32_t e = src % max_units;
    bool ok = r.status == AACG_PARSE_OK && e < r.n_units && unit_matches(u.d, p, refuse_pns);
    if (want) {
        /* with a map a frame is taken or refused as a whole: every lane of the frame looks at all of its elements (the plan lists
         * them next to each other, in the frame's order) and comes to the same answer; the first one reports it */
        ok = r.status == AACG_PARSE_OK && r.n_units == want;
        for (uint32_t k = 0; k < listed && ok; k++) ok = unit_matches(units[i - e + k].d, parsed[src - e + k], refuse_pns);
        if (!ok && e == 0 && r.status == AACG_PARSE_OK) results[src / max_units].status = AACG_PARSE_LAYOUT;
    }
    /* A refused frame's record (and the slots e >= n_units) is unspecified: nothing is taken from it.  Its unit keeps
     * the offsets the planner validated — a silent unit still loads its blocks (quant_load reads unconditionally). */
    u.d.tns_offset = 0;
    u.gmap[0] = u.gmap[1] = 0;
    if (ok) {
        u.d.flags = p.flags;
        for (int c = 0; c < 2; c++) {
            u.d.ch[c] = p.ch[c];
            u.d.ch[c].flags &= (uint8_t)~AACG_CHAN_TNS_PRESENT;                  /* no TNS records on this path */
            if (c < p.n_ch && p.ch[c].window_sequence == AACG_EIGHT_SHORT_SEQUENCE) {
                uint32_t gmap = 0;
                int w = 0;
                for (int g = 0; g < p.ch[c].group_count; g++)
                    for (int k = 0; k < p.ch[c].group_len[g] && w < 8; k++, w++) gmap |= (uint32_t)g << (4 * w);
                u.gmap[c] = gmap;
            }
        }
    } else {
        u.d.flags = 0;
        for (int c = 0; c < 2; c++) {
            aacg_chan_info z = {};
            z.group_count = 1; z.group_len[0] = 1;                              /* ONLY_LONG, sine, nothing coded */
            u.d.ch[c] = z;
        }
        if (!want || e == 0) atomicAdd(refused, 1u);    /* with a map: refused FRAMES (every element of one is refused together or for its own sake; the first counts) */
    }
    units[i] = u;
}

void aacg_refresh_launch(aacg_dev_unit* units, const aacg_unit_desc* parsed, aacg_parse_result* results, const aacg_refresh_map* map, uint32_t n_units,
                         uint32_t max_units, int refuse_pns, uint32_t* refused, hipStream_t s)
{
    hipLaunchKernelGGL(aacg_units_refresh, dim3((n_units + 255u) / 256u), dim3(256), 0, s, units, parsed, results, map, n_units, max_units, refuse_pns, refused);
}
