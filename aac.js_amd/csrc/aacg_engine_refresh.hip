/*
 * aacg_engine_refresh.hip — a plan's unit records rewritten on the device from what aacg_parse_device produced
 * (aacg_plan_refresh_from_parse, include/aacgpu.h).  The run tables of a plan only depend on which streams bring
 * how many frames of which element layout; consecutive batches of the same streams share them.  What changes from
 * batch to batch is what the parser found per frame — window sequence and shape, max_sfb, grouping, flags — and that
 * never has to visit the host: one lane per unit copies it from the parser's record into the plan's device record,
 * keeps the planner's part (stream, PCM offset, channel layout), derives the group-of-window map, and turns a frame
 * the parser refused into a silent one (ONLY_LONG, max_sfb 0), counting it.
 */
#include <hip/hip_runtime.h>

#include "aacg_device.h"

extern "C" __global__ __launch_bounds__(256)
void aacg_units_refresh(aacg_dev_unit* units, const aacg_unit_desc* parsed, const aacg_parse_result* results, uint32_t n_units,
                        uint32_t max_units, int refuse_pns, uint32_t* refused)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_units) return;
    aacg_dev_unit u = units[i];
    const aacg_unit_desc p = parsed[i];
    const aacg_parse_result r = results[i / max_units];
    const uint32_t e = i % max_units;
    /* (the blocks: the plan's run tables carry copies of the offsets, aacg_run.wave_coef — a frame the parser put elsewhere is not
     * the frame the plan was made for) */
    bool ok = r.status == AACG_PARSE_OK && e < r.n_units && p.n_ch == u.d.n_ch && p.channel == u.d.channel &&
              p.coef_offset == u.d.coef_offset && p.meta_offset == u.d.meta_offset;
    if (ok && refuse_pns && (p.flags & AACG_UNIT_HAS_PNS)) ok = false;      /* AACG_PNS_REFERENCE engines do not decode noise bands */
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
        atomicAdd(refused, 1u);
    }
    units[i] = u;
}

void aacg_refresh_launch(aacg_dev_unit* units, const aacg_unit_desc* parsed, const aacg_parse_result* results, uint32_t n_units,
                         uint32_t max_units, int refuse_pns, uint32_t* refused, hipStream_t s)
{
    hipLaunchKernelGGL(aacg_units_refresh, dim3((n_units + 255u) / 256u), dim3(256), 0, s, units, parsed, results, n_units, max_units, refuse_pns, refused);
}
