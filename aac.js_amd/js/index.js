/*
 * aac.js_amd/js — JavaScript host of the MI355X AAC-LC synthesis engine.
 *
 * Keeps the serial work where aac.js has it (ADTS demux, raw_data_block parse, Huffman: CPU,
 * JavaScript) and hands batches of parsed elements across the N-API layer (../napi) to the HIP
 * kernels, while presenting the Aurora.js decoder plugin surface of the reference
 * (src/decoder.js:29-216): init(), setCookie(buffer), readChunk() -> interleaved Float32Array.
 *
 * Own code throughout; nothing is imported from the reference.
 */
'use strict';
const path = require('path');

const INPUT_SPEC_F32 = 0, INPUT_QUANT_I16 = 1;
const OUTPUT_F32 = 0, OUTPUT_I16 = 1;
const CCE_REFERENCE = 0, CCE_SPEC = 1;          // GpuAACDecoder({ cceMode }), Engine({ cceMode })       // Engine({ outputKind }): OUTPUT_I16 engines take an Int16Array for pcm
const UNIT_BYTES = 64, META_WORDS = 120, FRAME = 1024, TNS_BYTES = 424, TNS_MAX_ORDER = 12;
const TNS_REFERENCE = 0, TNS_SPEC = 1, PNS_REFERENCE = 0, PNS_SPEC = 1;
const SAMPLE_RATES = [96000, 88200, 64000, 48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000, 7350];

let addon = null;
function loadAddon() {
    if (addon) return addon;
    addon = require(path.join(__dirname, '..', 'napi', 'aacgpu_napi.node'));
    // fails loudly if the HIP library was not built: there is no CPU fallback in the product path
    addon.load(process.env.AACGPU_LIB || path.join(__dirname, '..', 'csrc', 'libaacgpu.so'));
    return addon;
}

/* ---- unit table packing: mirrors aacg_unit_desc / aacg_band_meta in include/aacgpu.h ---------- */
function packChanInfo(view, off, ch) {
    view.setUint8(off + 0, ch.windowSequence);
    view.setUint8(off + 1, ch.windowShape);
    view.setUint8(off + 2, ch.windowShapePrev | 0);   // aac.js always has 0 here (fresh ICSInfo per frame)
    view.setUint8(off + 3, ch.maxSFB);
    view.setUint8(off + 4, ch.groupLength.length);
    view.setUint8(off + 5, ch.tns ? 1 : 0);           // AACG_CHAN_TNS_PRESENT (only read by TNS_SPEC engines)
    for (let g = 0; g < ch.groupLength.length; g++) view.setUint8(off + 8 + g, ch.groupLength[g]);
}

/* units: [{stream, pcmOffset, channel, nOutCh, coefOffset, metaOffset, commonWindow, maskPresent, ch: [info, info?]}] */
function packUnits(units) {
    const buf = new ArrayBuffer(UNIT_BYTES * units.length), view = new DataView(buf);
    units.forEach(function (u, i) {
        const o = UNIT_BYTES * i;
        view.setUint32(o + 0, u.stream, true);
        view.setUint32(o + 4, u.pcmOffset, true);
        view.setUint16(o + 8, u.channel, true);
        view.setUint16(o + 10, u.nOutCh, true);
        view.setUint8(o + 12, u.ch.length);
        view.setUint8(o + 13, (u.commonWindow ? 1 : 0) | (u.maskPresent ? 2 : 0) | (u.hasPns ? 4 : 0) | (u.cce ? 8 : 0));   // AACG_UNIT_*
        view.setUint32(o + 16, u.coefOffset, true);
        view.setUint32(o + 20, u.metaOffset, true);
        packChanInfo(view, o + 24, u.ch[0]);
        if (u.ch.length > 1) packChanInfo(view, o + 40, u.ch[1]);
        view.setUint32(o + 56, u.tnsOffset >>> 0, true);
        view.setUint32(o + 60, u.cceOffset >>> 0, true);     // reserved1: index of the coupling element's aacg_cce_info
    });
    return new Uint8Array(buf);
}

/* coupling side info -> aacg_cce_info records (include/aacgpu.h): [{ couplingPoint, targets: [{channel, gainList}],
 * gains: [Float32Array(120)...] }] */
const CCE_BYTES = 7716, CCE_MAX_TARGETS = 16;
function packCce(list) {
    const buf = new ArrayBuffer(CCE_BYTES * Math.max(1, list.length)), view = new DataView(buf);
    list.forEach(function (c, i) {
        const o = CCE_BYTES * i;
        if (c.targets.length > CCE_MAX_TARGETS || c.gains.length > CCE_MAX_TARGETS) throw new Error('coupling element with more than 16 targets');
        view.setUint8(o, c.couplingPoint); view.setUint8(o + 1, c.targets.length);
        c.targets.forEach(function (t, k) { view.setUint8(o + 4 + 2 * k, t.channel); view.setUint8(o + 5 + 2 * k, t.gainList); });
        c.gains.forEach(function (g, l) { for (let b = 0; b < META_WORDS; b++) view.setFloat32(o + 36 + 4 * (META_WORDS * l + b), g[b], true); });
    });
    return new Uint8Array(buf);
}

/* TNS side info -> aacg_tns_info records.  `list` holds one entry per channel slot, in the shape of the
 * reference's TNS object (tns.js:22-44: nFilt[w], length[w][filt], order[w][filt], direction[w][filt],
 * coef[w][filt][i]) plus `short` (EIGHT_SHORT_SEQUENCE), or null for a channel without TNS.  Long windows
 * put filter f of window 0 in slot f, short windows put window w's single filter in slot w. */
function packTns(list) {
    const buf = new ArrayBuffer(TNS_BYTES * Math.max(1, list.length)), view = new DataView(buf);
    list.forEach(function (t, i) {
        if (!t) return;
        const o = TNS_BYTES * i, nWin = t.short ? 8 : 1;
        for (let w = 0; w < nWin; w++) {
            const nf = t.nFilt[w] | 0;
            if (nf > (t.short ? 1 : 3)) throw new Error('TNS filter count out of range: ' + nf);
            view.setUint8(o + w, nf);
            for (let f = 0; f < nf; f++) {
                const fo = o + 8 + 52 * (t.short ? w : f), order = t.order[w][f] | 0;
                if (order > TNS_MAX_ORDER) throw new Error('TNS filter out of range: ' + order);   // tns.js:84-85 allows 20
                view.setUint8(fo, t.length[w][f]);
                view.setUint8(fo + 1, order);
                view.setUint8(fo + 2, t.direction[w][f] ? 1 : 0);
                for (let k = 0; k < order; k++) view.setFloat32(fo + 4 + 4 * k, t.coef[w][f][k], true);
            }
        }
    });
    return new Uint8Array(buf);
}

/* one aacg_tns_info record back into the shape of the reference's TNS object (the inverse of packTns) */
function unpackTns(bytes, index, short) {
    const view = new DataView(bytes.buffer, bytes.byteOffset + TNS_BYTES * index, TNS_BYTES), nWin = short ? 8 : 1;
    const t = { short: short, nFilt: [], length: [], order: [], direction: [], coef: [] };
    for (let w = 0; w < nWin; w++) {
        const nf = view.getUint8(w);
        t.nFilt.push(nf); t.length.push([]); t.order.push([]); t.direction.push([]); t.coef.push([]);
        for (let f = 0; f < nf; f++) {
            const fo = 8 + 52 * (short ? w : f), order = view.getUint8(fo + 1), coef = new Float32Array(order);
            for (let k = 0; k < order; k++) coef[k] = view.getFloat32(fo + 4 + 4 * k, true);
            t.length[w].push(view.getUint8(fo)); t.order[w].push(order); t.direction[w].push(!!view.getUint8(fo + 2)); t.coef[w].push(coef);
        }
    }
    return t;
}

function unpackUnits(bytes) {
    const view = new DataView(bytes.buffer, bytes.byteOffset, bytes.byteLength), out = [];
    function chan(off) {
        const n = view.getUint8(off + 4), gl = [];
        for (let g = 0; g < n; g++) gl.push(view.getUint8(off + 8 + g));
        return { windowSequence: view.getUint8(off), windowShape: view.getUint8(off + 1), windowShapePrev: view.getUint8(off + 2),
                 maxSFB: view.getUint8(off + 3), groupLength: gl, tnsPresent: !!(view.getUint8(off + 5) & 1) };
    }
    for (let o = 0; o < bytes.byteLength; o += UNIT_BYTES) {
        const nCh = view.getUint8(o + 12), flags = view.getUint8(o + 13);
        const u = { stream: view.getUint32(o, true), pcmOffset: view.getUint32(o + 4, true), channel: view.getUint16(o + 8, true),
                    nOutCh: view.getUint16(o + 10, true), coefOffset: view.getUint32(o + 16, true), metaOffset: view.getUint32(o + 20, true),
                    commonWindow: !!(flags & 1), maskPresent: !!(flags & 2), hasPns: !!(flags & 4), ch: [chan(o + 24)],
                    tnsOffset: view.getUint32(o + 56, true), tag: view.getUint16(o + 14, true) };
        if (nCh > 1) u.ch.push(chan(o + 40));
        out.push(u);
    }
    return out;
}

/* one 16-bit word per (group, sfb): sf index | negate<<9 | ms_used<<10 | bandType<<12 */
function packBandWord(bandType, sfIndex, negate, msUsed) {
    return (sfIndex & 0x1ff) | (negate ? 0x200 : 0) | (msUsed ? 0x400 : 0) | ((bandType & 15) << 12);
}

/* Pulse data (ics.js:175-201 parses it; ics.js:263-265 then throws 'TODO: add pulse data').  The standard's rule
 * (ISO/IEC 14496-3 4.6.3.3) works on the quantised integers, i.e. on this seam's input: coefficient
 * pulseOffset[i] moves away from zero by pulseAmp[i] (long windows only).  q: the channel's Int16Array(1024). */
function applyPulses(q, pulseOffset, pulseAmp) {
    for (let i = 0; i < pulseOffset.length; i++) {
        const k = pulseOffset[i];
        if (k > 1023) throw new Error('Pulse offset out of range: ' + k);       // ics.js:192-198
        q[k] += q[k] > 0 ? pulseAmp[i] : -pulseAmp[i];
    }
}

/* ---- engine wrapper ---------------------------------------------------------------------------- */
function Engine(opts) {
    opts = opts || {};
    this.addon = loadAddon();
    this.inputKind = opts.inputKind === undefined ? INPUT_QUANT_I16 : opts.inputKind;
    this.handle = this.addon.create({ deviceOrdinal: opts.deviceOrdinal | 0, sampleIndex: opts.sampleIndex === undefined ? 3 : opts.sampleIndex,
                                      maxStreams: opts.maxStreams || 1, maxChannels: opts.maxChannels || 2,
                                      maxBatchUnits: opts.maxBatchUnits | 0, inputKind: this.inputKind,
                                      tnsMode: opts.tnsMode | 0, pnsMode: opts.pnsMode | 0, outputKind: opts.outputKind | 0, cceMode: opts.cceMode | 0 });
}
/* tns: packTns(...) records for TNS_SPEC engines, else omitted */
/* cce: packCce(...) records for CCE_SPEC engines, else omitted */
Engine.prototype.decodeBatch = function (units, coeffs, meta, pcm, tns, cce) {
    return this.addon.decodeBatch(this.handle, units, coeffs, meta || null, pcm, tns || null, cce || null);   // throws on error
};
/* the same off the JavaScript thread: resolves with `pcm`; one call in flight per engine (the kernels chain
 * through the overlap state, so batches of one engine are decoded in order) */
Engine.prototype.decodeBatchAsync = function (units, coeffs, meta, pcm, tns) {
    return this.addon.decodeBatchAsync(this.handle, units, coeffs, meta || null, pcm, tns || null);
};
Engine.prototype.resetStream = function (s) { this.addon.resetStream(this.handle, s); };
Engine.prototype.getOverlap = function (s, c) { return this.addon.getOverlap(this.handle, s, c, new Float32Array(FRAME)); };
Engine.prototype.setOverlap = function (s, c, v) { this.addon.setOverlap(this.handle, s, c, v); };

/* ---- minimal MSB-first bit reader (for setCookie; the frame parser brings its own stream) ------- */
function BitReader(bytes) { this.bytes = bytes; this.pos = 0; }
BitReader.prototype.read = function (n) {
    let v = 0;
    for (let i = 0; i < n; i++, this.pos++) {
        if ((this.pos >> 3) >= this.bytes.length) throw new Error('Insufficient data');
        v = (v * 2) + ((this.bytes[this.pos >> 3] >> (7 - (this.pos & 7))) & 1);
    }
    return v;
};
BitReader.prototype.advance = function (n) { this.pos += n; };

/*
 * GpuAACDecoder — the plugin surface of aac.js's AACDecoder (src/decoder.js:49-216) on top of the engine.
 *
 *   init()                 format.floatingPoint = true                       (decoder.js:49-51)
 *   setCookie(buffer)      AudioSpecificConfig -> this.config, channelsPerFrame; allocates the stream's
 *                          filterbank state = resets the engine's stream slot  (decoder.js:53-113)
 *   readChunk()            one frame of interleaved Float32Array(1024 * channels) in [-1, 1), or throws
 *
 * The serial bitstream parse is delegated to `frontend.parseFrame(decoder)`, which must return
 *   { elements: [{ type: 'sce'|'lfe'|'cpe', commonWindow, maskPresent, ch: [chanInfo...] }],
 *     q: Int16Array(1024 * nCh), meta: Uint16Array(120 * nCh) }        (channel order = element order)
 * or null when no complete frame is buffered.  To batch without changing the surface, readChunk parses
 * ahead every complete frame already buffered (up to `lookahead`), submits ONE batch, returns frame 0 and
 * serves the rest from a queue; overlap state is only advanced by frames that parsed completely, as in
 * the reference where process() runs after align() (decoder.js:200-201).
 */
function GpuAACDecoder(opts) {
    opts = opts || {};
    this.format = opts.format || {};
    this.frontend = opts.frontend || null;
    this.engine = opts.engine || null;
    this.stream = opts.stream | 0;           // engine stream slot of this decoder instance
    /* a SharedEngine (shared_engine.js): this decoder takes a stream slot of an engine it shares with the other decoders
     * registered there, and a look-ahead batch holds the buffered frames of ALL of them — one launch instead of one per
     * stream (decoder.js:125-216 is one instance per stream; the GPU wants their frames together) */
    this.shared = opts.shared || null;
    this.lookahead = opts.lookahead || 16;
    /* aac.js always windows the first half of a frame with the SINE shape, because it builds a fresh ICSInfo
     * per frame and so loses windowShape[0] (decoder.js:145,153; SURVEY.md 9.1).  false (default) reproduces
     * that; true carries the previous frame's shape per channel as the standard prescribes. */
    this.carryWindowShape = !!opts.carryWindowShape;
    /* aac.js's TNS.process leaves the spectrum untouched (NaN loop bounds, tns.js:106,122).  TNS_REFERENCE
     * (default) reproduces that; TNS_SPEC applies the filter, taking each channel's side info from
     * chanInfo.tns (an object shaped like the reference's TNS instance, tns.js:22-44). */
    this.tnsMode = opts.tnsMode | 0;
    /* PNS_REFERENCE (default): a frame with NOISE_BT bands (element.hasPns) is refused, aac.js produces NaN there
     * (ics.js:234,239); PNS_SPEC fills the bands as ics.js:228-243 was meant to */
    this.pnsMode = opts.pnsMode | 0;
    /* CCE_REFERENCE (default): coupling channel elements are parsed and dropped, as aac.js effectively does (decoder.js:418
     * never matches).  CCE_SPEC: a front end that keeps them ({ coupling: true, referenceQuirks: false }) hands them on, and
     * the engine applies them; maxCoupling = coupling elements a frame may carry (each needs a stream channel of its own). */
    this.cceMode = opts.cceMode | 0;
    this.maxCoupling = this.cceMode === CCE_SPEC ? (opts.maxCoupling || 1) : 0;
    /* chanInfo.pulse = { offset: [...], amp: [...] } (ics.pulseOffset / pulseAmp): aac.js throws 'TODO: add pulse
     * data' on such a frame (ics.js:263-265); applyPulses: true adds them to the quantised spectrum instead. */
    this.applyPulses = !!opts.applyPulses;
    this.prevShape = [];
    this.queue = [];
}
GpuAACDecoder.prototype.init = function () { this.format.floatingPoint = true; };

GpuAACDecoder.prototype.setCookie = function (buffer) {
    const bytes = buffer.data || buffer, s = new BitReader(bytes), cfg = this.config = {};
    cfg.profile = s.read(5);
    if (cfg.profile === 31) cfg.profile = 32 + s.read(6);
    cfg.sampleIndex = s.read(4);
    if (cfg.sampleIndex === 0x0f) {
        cfg.sampleRate = s.read(24);
        const i = SAMPLE_RATES.indexOf(cfg.sampleRate);
        if (i >= 0) cfg.sampleIndex = i;
    } else {
        cfg.sampleRate = SAMPLE_RATES[cfg.sampleIndex];
    }
    cfg.chanConfig = s.read(4);
    this.format.channelsPerFrame = cfg.chanConfig;
    if (cfg.profile !== 1 && cfg.profile !== 2 && cfg.profile !== 4)
        throw new Error('AAC profile ' + cfg.profile + ' not supported.');
    if (s.read(1)) throw new Error('frameLengthFlag not supported');
    cfg.frameLength = FRAME;
    if (s.read(1)) s.advance(14);
    if (s.read(1)) { if (cfg.profile > 16) s.advance(3); s.advance(1); }
    if (cfg.chanConfig === 0) throw new Error('PCE unimplemented');
    if (this.shared) this.shared.attach(this);          // engine + stream slot for this sample rate
    if (!this.engine)
        this.engine = new Engine({ sampleIndex: cfg.sampleIndex, maxStreams: this.stream + 1, maxChannels: cfg.chanConfig + this.maxCoupling, inputKind: INPUT_QUANT_I16,
                                   tnsMode: this.tnsMode, pnsMode: this.pnsMode, cceMode: this.cceMode });
    this.engine.resetStream(this.stream);    // new FilterBank(false, chanConfig): zeroed overlaps (filter_bank.js:38-41)
};

/* elements of one parsed frame -> unit records; channel indices assigned in element order, elements beyond
 * chanConfig channels dropped (decoder.js:233-247) */
GpuAACDecoder.prototype.unitsOfFrame = function (frame, frameSlot, blockBase, tnsList, cceList, pcmBase) {
    const C = this.config.chanConfig, units = [];
    pcmBase = pcmBase | 0;                               // float offset of this decoder's frames in the batch's PCM buffer
    let channel = 0, block = blockBase;
    const audio = frame.elements.filter(function (e) { return e.type !== 'cce'; });
    const coupling = frame.elements.filter(function (e) { return e.type === 'cce'; });
    if (coupling.length && !cceList) throw new Error('coupling channel element in the frame: the decoder was not created with cceMode: CCE_SPEC');
    if (coupling.length > this.maxCoupling) throw new Error('more coupling channel elements in a frame than maxCoupling allows for');
    /* everything that can refuse the frame comes first: a frame that throws leaves no trace in the decoder's state */
    for (const e of audio) {
        if (channel >= C) break;
        if (e.gainPresent) throw new Error('Gain control not implemented');
        if (e.hasPns && this.pnsMode !== PNS_SPEC) throw new Error('aacgpu: NOISE_BT (PNS) band: not decodable by the reference either (pnsMode: PNS_SPEC fills them)');
        for (const ch of e.ch) if (ch.pulse && !this.applyPulses) throw new Error('TODO: add pulse data');
        channel += e.type === 'cpe' ? 2 : 1;
    }
    channel = 0;
    /* blocks follow the order of frame.elements (that is how the front end lays q / meta out) */
    const blockOf = new Map();
    { let b = blockBase; for (const e of frame.elements) { blockOf.set(e, b); b += e.type === 'cpe' ? 2 : 1; } }
    const placed = [];                                   // audio elements with their first output channel (decoder.js:233-247)
    for (const e of audio) {
        const n = e.type === 'cpe' ? 2 : 1;
        if (channel >= C) break;
        block = blockOf.get(e);
        placed.push({ e: e, channel: channel });
        for (let c = 0; c < n; c++) {
            e.ch[c].windowShapePrev = this.carryWindowShape ? (this.prevShape[channel + c] | 0) : 0;
            this.prevShape[channel + c] = e.ch[c].windowShape;
            if (e.ch[c].pulse) {
                applyPulses(frame.q.subarray(FRAME * (block - blockBase + c), FRAME * (block - blockBase + c + 1)), e.ch[c].pulse.offset, e.ch[c].pulse.amp);
            }
        }
        let tnsOffset = 0;
        if (tnsList && e.ch.some(function (c) { return !!c.tns; })) {       // ics.tnsPresent (ics.js:71), TNS_SPEC engines only
            tnsOffset = tnsList.length;
            for (let c = 0; c < n; c++) {
                if (e.ch[c].tns) e.ch[c].tns.short = e.ch[c].windowSequence === 2;
                tnsList.push(e.ch[c].tns || null);
            }
        }
        units.push({ stream: this.stream, pcmOffset: pcmBase + frameSlot * FRAME * C, channel: channel, nOutCh: C, coefOffset: block, metaOffset: block,
                     commonWindow: !!e.commonWindow, maskPresent: !!e.maskPresent, hasPns: !!e.hasPns, ch: e.ch, tnsOffset: tnsOffset });
        channel += n;
    }
    /* coupling channel elements (CCE_SPEC): stream channel C + k for the k-th of the frame; which output channels an
     * (is-a-pair, id, select) triple means is decided as decoder.js:411-431 walks them — with the comparison of
     * channelPair that can be true (decoder.js:418 compares a number with a boolean) and every coupled element visited
     * (decoder.js:416 stops one short of cce.js:53) */
    coupling.forEach((e, k) => {
        const targets = [];
        let index = 0;
        for (const t of e.targets) {
            const hit = placed.find(function (p) { return (p.e.type === 'cpe') === t.pair && p.e.id === t.id; });
            if (hit) {
                if (t.sel !== 1) { targets.push({ channel: hit.channel, gainList: index }); if (t.sel) index++; }
                if (t.sel !== 2) targets.push({ channel: hit.channel + 1, gainList: index++ });
            } else index += 1 + (t.sel === 3 ? 1 : 0);
        }
        e.ch[0].windowShapePrev = this.carryWindowShape ? (this.prevShape[C + k] | 0) : 0;
        this.prevShape[C + k] = e.ch[0].windowShape;
        const info = Object.assign({}, e.ch[0], { tns: null });       // the reference never runs a coupling element's own TNS either
        units.push({ stream: this.stream, pcmOffset: pcmBase + frameSlot * FRAME * C, channel: C + k, nOutCh: C, coefOffset: blockOf.get(e), metaOffset: blockOf.get(e),
                     commonWindow: false, maskPresent: false, hasPns: false, cce: true, cceOffset: cceList.length, ch: [info], tnsOffset: 0 });
        cceList.push({ couplingPoint: e.couplingPoint, targets: targets, gains: e.gains });
    });
    return units;
};

/* The queue holds, in stream order, the PCM of frames decoded ahead and — as an Error — a frame the look-ahead found
 * malformed: it is thrown by the readChunk call that reaches it, after the good frames before it have been returned,
 * which is what a caller of the reference sees frame by frame (decoder.js:125-201 throws at exactly that frame). */
GpuAACDecoder.prototype.readChunk = function () {
    if (!this.queue.length) { if (this.shared && this.group) this.shared.flush(); else this.decodeAhead(); }
    if (!this.queue.length) return null;
    const next = this.queue.shift();
    if (next instanceof Error) throw next;
    return next;
};

/* Parse ahead every complete frame already buffered (up to `lookahead`): { frames, units, nBlocks, failed }.  The units'
 * block and PCM offsets count from (blockBase, pcmBase) — a batch of its own starts at (0, 0), a SharedEngine batch wherever the
 * decoders before this one ended.  Nothing is decoded here. */
GpuAACDecoder.prototype.collectAhead = function (blockBase, pcmBase, tnsList, cceList) {
    if (this.config.profile === 1) throw new Error('Main prediction unimplemented');
    if (this.config.profile === 4) throw new Error('LTP prediction unimplemented');
    const frames = [];
    let units = [], block = blockBase, failed = null, layout = null;
    /* the elements that carry state from frame to frame (overlap-add): which ones a frame has, in order.  A batch is one
     * chain per element for the planner — an element (a coupling element with its own filterbank included) that appears
     * or disappears ends the batch; the frame that differs starts the next one (it waits in this.pendingFrame). */
    const layoutOf = function (f) {
        let k = 0;                                      // coupling elements count from stream channel C on, in frame order
        return f.elements.map(function (e) {
            if (e.couplingPoint === undefined) return e.type;          // channels are dealt out in element order (decoder.js:233-247): the tags do not matter
            return e.couplingPoint === 2 ? 'cce@' + k++ : (k++, '');     // dependent coupling (spectral domain) carries no state
        }).join(',');
    };
    while (frames.length < this.lookahead) {
        /* both front ends have consumed a frame by the time they throw for it, and return null on underflow: an
         * exception here is a malformed (or unsupported) frame, kept in order behind the frames parsed so far */
        try {
            const f = this.pendingFrame || this.frontend.parseFrame(this);
            this.pendingFrame = null;
            if (!f) break;
            const sig = layoutOf(f);
            if (layout !== null && sig !== layout) { this.pendingFrame = f; break; }
            layout = sig;
            units = units.concat(this.unitsOfFrame(f, frames.length, block, tnsList, cceList, pcmBase));
            block += f.q.length / FRAME;
            frames.push(f);
        } catch (err) { failed = err instanceof Error ? err : new Error(String(err)); break; }
    }
    return { frames: frames, units: units, nBlocks: block - blockBase, failed: failed };
};

/* the frames' spectra and band words into the batch's arrays, from block `blockBase` on */
GpuAACDecoder.prototype.fillBatch = function (part, blockBase, q, meta) {
    let b = blockBase;
    for (const f of part.frames) { q.set(f.q, b * FRAME); meta.set(f.meta, b * META_WORDS); b += f.q.length / FRAME; }
};

/* what a batch produced, into the queue in stream order: the frames' PCM (each its own array: the caller owns it), or the
 * engine's refusal in their place; then the parse error found behind them, if any */
GpuAACDecoder.prototype.deliver = function (part, pcm, pcmBase, refused) {
    const C = this.config.chanConfig;
    if (part.frames.length) {
        if (refused) this.queue.push(refused);
        else for (let i = 0; i < part.frames.length; i++) this.queue.push(pcm.slice(pcmBase + i * FRAME * C, pcmBase + (i + 1) * FRAME * C));
    }
    if (part.failed) this.queue.push(part.failed);
};

GpuAACDecoder.prototype.decodeAhead = function () {
    const C = this.config.chanConfig, tnsList = this.tnsMode === TNS_SPEC ? [] : null, cceList = this.cceMode === CCE_SPEC ? [] : null;
    const part = this.collectAhead(0, 0, tnsList, cceList);
    let pcm = null, refused = null;
    if (part.frames.length) {
        const q = new Int16Array(part.nBlocks * FRAME), meta = new Uint16Array(part.nBlocks * META_WORDS);
        this.fillBatch(part, 0, q, meta);
        pcm = new Float32Array(part.frames.length * FRAME * C);
        /* an engine error costs this batch's frames, not the order: it is queued where their PCM would have been,
         * in front of a parse error found behind them, and the next call goes on with the frames that follow */
        try {
            this.engine.decodeBatch(packUnits(part.units), q, meta, pcm, tnsList && tnsList.length ? packTns(tnsList) : null,
                                    cceList && cceList.length ? packCce(cceList) : null);
        } catch (err) { refused = err instanceof Error ? err : new Error(String(err)); }
    }
    this.deliver(part, pcm, 0, refused);
};

/* a decoder that is done with its stream gives its slot of a SharedEngine back */
GpuAACDecoder.prototype.close = function () { if (this.shared && this.group) this.shared.detach(this); };

/* bytes from the demuxer ('data' events of AdtsDemuxer / an MP4 demuxer's samples) to the front end */
GpuAACDecoder.prototype.feed = function (bytes) {
    bytes = bytes.data || bytes;
    if (this.resident) {                                // a resident SharedEngine parses on the device: the decoder only keeps the bytes
        const rest = this.rbuf.length - this.rpos;
        if (!rest) { this.rbuf = bytes; this.rpos = 0; return; }
        const joined = new Uint8Array(rest + bytes.length);
        joined.set(this.rbuf.subarray(this.rpos)); joined.set(bytes, rest);
        this.rbuf = joined; this.rpos = 0;
        return;
    }
    this.frontend.push(bytes);
};
GpuAACDecoder.prototype.feedPacket = function (bytes, multi) { this.frontend.pushPacket(bytes.data || bytes, multi); };   // multi: the buffer may hold several samples

module.exports = { Engine, GpuAACDecoder, BitReader, packUnits, unpackUnits, packBandWord, packTns, unpackTns, packCce, CCE_REFERENCE, CCE_SPEC, CCE_BYTES, applyPulses, loadAddon,
                   INPUT_SPEC_F32, INPUT_QUANT_I16, OUTPUT_F32, OUTPUT_I16, TNS_REFERENCE, TNS_SPEC, PNS_REFERENCE, PNS_SPEC, UNIT_BYTES, META_WORDS, TNS_BYTES, SAMPLE_RATES };
/* the bitstream front end and its pieces (loaded on first use: they require this module themselves) */
for (const [name, file] of [['SharedEngine', './shared_engine.js'], ['FrontEnd', './frontend.js'], ['GpuFrontEnd', './gpu_frontend.js'], ['codebooks', './codebooks.js'], ['adts', './adts.js'], ['BitStream', './bits.js']])
    Object.defineProperty(module.exports, name, { enumerable: true, get: function () { const m = require(file); return m[name] || m; } });
