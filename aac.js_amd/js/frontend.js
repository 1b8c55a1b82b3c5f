/*
 * aac.js_amd/js/frontend.js — the serial half of the decoder: raw_data_block -> what the engine takes.
 *
 * Does what the reference does between `stream.peek(12)` and `this.process(elements)` (decoder.js:126-201:
 * optional ADTS header, element loop, ics.js:56-201,279-314, cpe.js:37-75, tns.js:68-103, cce.js:45-119,
 * Huffman decode), with one difference in kind: nothing is dequantised here.  Each channel leaves as
 *   q     Int16Array(1024)   the quantised integers in ICStream.data's index order (window w at w*128)
 *   meta  Uint16Array(120)   one band word per (group, sfb): scalefactor-table index, negate, ms_used, band type
 * and dequantisation, M/S, intensity, (TNS, PNS,) IMDCT, windowing and overlap-add run on the GPU
 * (AACG_INPUT_QUANT_I16, include/aacgpu.h).  The returned frame object is what GpuAACDecoder.readChunk consumes
 * (index.js):  { elements: [{ type, id, commonWindow, maskPresent, hasPns, ch: [chanInfo, chanInfo?] }], q, meta }
 * with chanInfo = { windowSequence, windowShape, maxSFB, groupLength, tns?, pulse? }.
 *
 * Error behaviour follows the reference message for message where it has one ('Invalid band type: 12',
 * 'Too many bands', 'Scalefactor out of range', 'TODO: PCE_ELEMENT', 'Prediction not implemented.', ...).
 * Coupling channel elements are parsed and dropped, as the reference does (decoder.js:162-166 collects them,
 * decoder.js:218-248 never applies them).
 *
 * Input: ADTS bytes through push() — frames are delimited by the header's frame_length, and a frame is only
 * parsed once it is complete, so parseFrame() returns null instead of failing half-way — or whole
 * raw_data_blocks (MP4 samples) through pushPacket().
 *
 * The Huffman code words come from ./codebooks.js (the standard's books ship in aac.js_amd/data/).
 * Own code; nothing is imported from the reference.
 */
'use strict';
const { BitStream } = require('./bits.js');
const codebooks = require('./codebooks.js');
const adts = require('./adts.js');

const FRAME = 1024, META_WORDS = 120;
const SCE = 0, CPE = 1, CCE = 2, LFE = 3, DSE = 4, PCE = 5, FIL = 6, END = 7;
const ZERO_BT = 0, NOISE_BT = 13, INTENSITY_BT2 = 14, INTENSITY_BT = 15;
const EIGHT_SHORT = 2;

/* scalefactor band widths as (width, repeat) runs per sampleIndex (ISO/IEC 14496-3 4.5.4; the reference lists the
 * resulting offsets, tables.js:60-160) */
const SWB_LONG = [
    [4, 14, 8, 5, 12, 5, 16, 2, 24, 1, 28, 1, 36, 1, 44, 1, 64, 11], null, [4, 14, 8, 4, 12, 3, 16, 3, 20, 1, 24, 2, 28, 1, 36, 1, 40, 18],
    [4, 10, 8, 7, 12, 4, 16, 2, 20, 2, 24, 2, 28, 2, 32, 19, 96, 1], null, [4, 10, 8, 7, 12, 4, 16, 2, 20, 2, 24, 2, 28, 2, 32, 22],
    [4, 11, 8, 10, 12, 4, 16, 3, 20, 2, 24, 2, 28, 2, 32, 1, 36, 2, 40, 1, 44, 1, 48, 1, 52, 2, 64, 5], null,
    [8, 11, 12, 9, 16, 4, 20, 3, 24, 2, 28, 2, 32, 1, 36, 1, 40, 2, 44, 1, 48, 1, 52, 1, 56, 1, 60, 1, 64, 3], null, null,
    [12, 13, 16, 7, 20, 4, 24, 3, 28, 2, 32, 1, 36, 2, 40, 1, 44, 1, 48, 1, 52, 1, 56, 1, 60, 1, 64, 1, 80, 1]];
const SWB_SHORT = [
    [4, 6, 8, 3, 16, 1, 28, 1, 36, 1], null, null, [4, 5, 8, 3, 12, 3, 16, 3], null, null, [4, 7, 8, 3, 12, 2, 16, 2, 20, 1], null,
    [4, 8, 8, 2, 12, 2, 16, 1, 20, 2], null, null, [4, 7, 8, 4, 12, 1, 16, 1, 20, 2]];
function expandRuns(table, sampleIndex) {
    let i = sampleIndex;
    while (table[i] === null) i--;                      // a null entry shares the layout of the rate above it
    const off = [0];
    for (let r = 0; r < table[i].length; r += 2)
        for (let k = 0; k < table[i][r + 1]; k++) off.push(off[off.length - 1] + table[i][r]);
    return Uint16Array.from(off);
}
function swbOffsets(sampleIndex, short) {
    if (!(sampleIndex >= 0 && sampleIndex <= 11)) throw new Error('sampleIndex out of range: ' + sampleIndex);
    return expandRuns(short ? SWB_SHORT : SWB_LONG, sampleIndex);
}

/* the values a `bits`-wide TNS coefficient field selects (ISO/IEC 14496-3 4.6.9.3: sin(q / iqfac) on a
 * resBits-bit grid, the field being q's low bits when coef_compress drops the top one; the sign convention is
 * the reference's tables, tns.js:50-61) */
function tnsCoefTable(resBits, compress) {
    const n = 1 << (resBits - compress), half = 1 << (resBits - 1), out = new Float32Array(n);
    for (let i = 0; i < n; i++) {
        const s = i >= n / 2 ? i - n : i;
        out[i] = -Math.sin(s / ((s >= 0 ? half - 0.5 : half + 0.5) / (Math.PI / 2)));
    }
    return out;
}
const TNS_TABLES = [tnsCoefTable(3, 0), tnsCoefTable(4, 0), tnsCoefTable(3, 1), tnsCoefTable(4, 1)];   // [2*compress + res]

function FrontEnd(opts) {
    opts = opts || {};
    this.cb = codebooks.load(opts);
    this.buf = new Uint8Array(0);         // ADTS bytes not yet consumed
    this.packets = [];                    // raw_data_blocks not yet consumed
    this.swb = {};
    this.referenceQuirks = opts.referenceQuirks !== false;   // coupling channel elements: see cce()
    /* coupling: true (with referenceQuirks: false, the standard's syntax): coupling channel elements are kept — their
     * spectrum, targets and gain lists — for decoders in CCE_SPEC mode; the default drops them like the reference */
    this.keepCoupling = !!opts.coupling && !this.referenceQuirks;
    this.scratchQ = new Int16Array(FRAME);
    this.scratchMeta = new Uint16Array(META_WORDS);
}

FrontEnd.prototype.push = function (bytes) {
    if (!this.buf.length) { this.buf = bytes; return; }
    const joined = new Uint8Array(this.buf.length + bytes.length);
    joined.set(this.buf); joined.set(bytes, this.buf.length);
    this.buf = joined;
};
/* pushPacket(bytes): one raw_data_block (an MP4 sample).  pushPacket(bytes, true): whatever an MP4 demuxer emitted — a
 * buffer that may hold several byte-aligned raw_data_blocks back to back (Aurora's M4A demuxer emits a chunk's contiguous
 * samples as one buffer); parseFrame() then goes on behind a block where the reference's bitstream would. */
FrontEnd.prototype.pushPacket = function (bytes, multi) { this.packets.push({ bytes: bytes, multi: !!multi }); };

FrontEnd.prototype.offsets = function (sampleIndex, short) {
    const key = sampleIndex * 2 + (short ? 1 : 0);
    return this.swb[key] || (this.swb[key] = swbOffsets(sampleIndex, short));
};

/* ---- ics_info (ics.js:279-314) ---------------------------------------------------------------------- */
FrontEnd.prototype.icsInfo = function (bits, config) {
    bits.advance(1);                                     // reserved
    const info = { windowSequence: bits.read(2), windowShape: bits.read(1), groupLength: [1] };
    info.short = info.windowSequence === EIGHT_SHORT;
    if (info.short) {
        info.maxSFB = bits.read(4);
        for (let i = 0; i < 7; i++) {
            if (bits.read1()) info.groupLength[info.groupLength.length - 1]++;
            else info.groupLength.push(1);
        }
    } else {
        info.maxSFB = bits.read(6);
        if (bits.read1()) throw new Error('Prediction not implemented.');
    }
    info.offsets = this.offsets(config.sampleIndex, info.short);
    info.swbCount = info.offsets.length - 1;
    if (info.maxSFB > info.swbCount) throw new Error('maxSFB out of range: ' + info.maxSFB + ' > ' + info.swbCount);   // the reference reads past its offset table here
    return info;
};

/* ---- individual_channel_stream (ics.js:56-201) ---------------------------------------------------------
 * q / meta: this channel's 1024 / 120 element views (zero-filled).  Returns the channel's chanInfo, with
 * .hasPns set if a NOISE_BT band occurred. */
FrontEnd.prototype.ics = function (bits, config, info, q, meta) {
    const cb = this.cb, globalGain = bits.read(8);
    if (!info) info = this.icsInfo(bits, config);
    const groups = info.groupLength.length, maxSFB = info.maxSFB, nb = groups * maxSFB;
    const bandType = new Uint8Array(nb);

    /* section_data (ics.js:83-116) */
    const lenBits = info.short ? 3 : 5, escape = (1 << lenBits) - 1;
    for (let g = 0, idx = 0; g < groups; g++) {
        for (let k = 0; k < maxSFB;) {
            const bt = bits.read(4);
            if (bt === 12) throw new Error('Invalid band type: 12');
            let end = k, incr;
            while ((incr = bits.read(lenBits)) === escape) end += incr;
            end += incr;
            if (end > maxSFB) throw new Error('Too many bands (' + end + ' > ' + maxSFB + ')');
            for (; k < end; k++, idx++) bandType[idx] = bt;
        }
    }

    /* scale_factor_data (ics.js:118-173): three running values — spectrum, noise, intensity */
    let sfSpec = globalGain, sfNoise = globalGain - 90, sfInt = 0, firstNoise = true, hasPns = false;
    for (let idx = 0; idx < nb; idx++) {
        const bt = bandType[idx];
        let word = bt << 12;
        if (bt === ZERO_BT) {
            /* nothing coded */
        } else if (bt === INTENSITY_BT || bt === INTENSITY_BT2) {
            sfInt += cb.scaleFactor(bits) - 60;
            word |= 200 - Math.min(Math.max(sfInt, -155), 100);
        } else if (bt === NOISE_BT) {
            if (firstNoise) { sfNoise += bits.read(9) - 256; firstNoise = false; }
            else sfNoise += cb.scaleFactor(bits) - 60;
            word |= (200 + Math.min(Math.max(sfNoise, -100), 155)) | 0x200;              // the band's gain is -SF (ics.js:159)
            hasPns = true;
        } else {
            sfSpec += cb.scaleFactor(bits) - 60;
            if (sfSpec > 255 || sfSpec < -100) throw new Error('Scalefactor out of range: ' + sfSpec);   // ics.js:164 has the upper test only
            word |= sfSpec + 100;
        }
        meta[idx] = word;
    }

    const chan = { windowSequence: info.windowSequence, windowShape: info.windowShape, maxSFB: maxSFB, groupLength: info.groupLength, hasPns: hasPns };

    /* pulse_data (ics.js:175-201) */
    if (bits.read1()) {
        if (info.short) throw new Error('Pulse tool not allowed in eight short sequence.');
        const count = bits.read(2) + 1, swb = bits.read(6);
        if (swb >= info.swbCount) throw new Error('Pulse SWB out of range: ' + swb);
        const offset = [], amp = [];
        for (let i = 0, at = info.offsets[swb]; i < count; i++) {
            at += bits.read(5);
            if (at > 1023) throw new Error('Pulse offset out of range: ' + at);
            offset.push(at); amp.push(bits.read(4));
        }
        chan.pulse = { offset: offset, amp: amp };
    }

    /* tns_data (tns.js:68-103), kept in the shape of the reference's TNS object */
    if (bits.read1()) chan.tns = this.tns(bits, info);

    if (bits.read1()) throw new Error('TODO: decode gain control/SSR');

    /* spectral_data (ics.js:203-261 without the dequantisation) */
    const off = info.offsets;
    for (let g = 0, idx = 0, groupOff = 0; g < groups; g++) {
        const glen = info.groupLength[g];
        for (let sfb = 0; sfb < maxSFB; sfb++, idx++) {
            const bt = bandType[idx];
            if (bt === ZERO_BT || bt >= NOISE_BT) continue;
            const step = bt < 5 ? 4 : 2, lo = groupOff + off[sfb], hi = groupOff + off[sfb + 1];
            for (let w = 0; w < glen; w++)
                for (let k = lo + w * 128; k < hi + w * 128; k += step) cb.spectral(bits, bt, q, k);
        }
        groupOff += glen * 128;
    }
    return chan;
};

FrontEnd.prototype.tns = function (bits, info) {
    const windows = info.short ? 8 : 1, nBits = info.short ? 1 : 2, lenBits = info.short ? 4 : 6, ordBits = info.short ? 3 : 5;
    const t = { nFilt: [], length: [], order: [], direction: [], coef: [] };
    for (let w = 0; w < windows; w++) {
        const nf = bits.read(nBits);
        t.nFilt.push(nf); t.length.push([]); t.order.push([]); t.direction.push([]); t.coef.push([]);
        if (!nf) continue;
        const res = bits.read1();
        for (let f = 0; f < nf; f++) {
            t.length[w].push(bits.read(lenBits));
            const order = bits.read(ordBits);
            if (order > 20) throw new Error('TNS filter out of range: ' + order);
            t.order[w].push(order);
            let direction = false;
            const coef = new Float32Array(order);
            if (order) {
                direction = !!bits.read1();
                const compress = bits.read1(), table = TNS_TABLES[2 * compress + res], width = res + 3 - compress;
                for (let i = 0; i < order; i++) coef[i] = table[bits.read(width)];
            }
            t.direction[w].push(direction); t.coef[w].push(coef);
        }
    }
    return t;
};

/* ---- channel_pair_element (cpe.js:37-75) ---------------------------------------------------------------- */
FrontEnd.prototype.cpe = function (bits, config, q, meta) {
    const e = { type: 'cpe', commonWindow: !!bits.read1(), maskPresent: false, hasPns: false, ch: [] };
    let info = null, ms = null;
    if (e.commonWindow) {
        info = this.icsInfo(bits, config);
        const mask = bits.read(2), nb = info.groupLength.length * info.maxSFB;
        e.maskPresent = mask !== 0;
        if (mask === 3) throw new Error('Reserved ms mask type: ' + mask);
        if (mask) {
            ms = new Uint8Array(nb);
            for (let i = 0; i < nb; i++) ms[i] = mask === 2 ? 1 : bits.read1();
        }
    }
    e.ch.push(this.ics(bits, config, info, q.subarray(0, FRAME), meta.subarray(0, META_WORDS)));
    e.ch.push(this.ics(bits, config, info, q.subarray(FRAME, 2 * FRAME), meta.subarray(META_WORDS, 2 * META_WORDS)));
    if (ms) for (let i = 0; i < ms.length; i++) if (ms[i]) meta[i] |= 0x400;             // ms_used lives in the left channel's words
    e.hasPns = e.ch[0].hasPns || e.ch[1].hasPns;
    return e;
};

/* ---- coupling_channel_element: consume exactly the bits the reference consumes (cce.js:45-119) ------------ */
const CCE_SCALE = [Math.pow(2, 1 / 8), Math.pow(2, 1 / 4), Math.SQRT2, 2];     // cce.js:37-42
FrontEnd.prototype.cce = function (bits, config, keep) {
    let point = 2 * bits.read1(), gains = 0;
    const coupled = bits.read(3), targets = [];
    for (let i = 0; i <= coupled; i++) {
        gains++;
        const pair = bits.read1(), id = bits.read(4), sel = pair ? bits.read(2) : 2;   // cce.js:56-65
        if (pair && sel === 3) gains++;
        targets.push({ pair: pair === 1, id: id, sel: sel });
    }
    point += bits.read1();
    point |= point >>> 1;
    const sign = bits.read1(), scale = CCE_SCALE[bits.read(2)];
    const q = keep ? new Int16Array(FRAME) : this.scratchQ, meta = keep ? new Uint16Array(META_WORDS) : this.scratchMeta;
    this.scratchQ.fill(0); this.scratchMeta.fill(0);
    const chan = this.ics(bits, config, null, q, meta), nb = chan.groupLength.length * chan.maxSFB;
    if (keep) {
        /* the standard's syntax (4.4.2.1 coupling_channel_element), gains as cce.js:77-107 forms them: list 0 is all
         * ones; a list is one common gain (cge) or one per coded band, coded differentially with the sign in the low
         * bit unless gain_element_sign says otherwise.  Here a list is indexed by the band like bandTypes (the
         * reference packs it densely, cce.js:103, and then reads it by band, cce.js:146: see oracle/aac_oracle.c). */
        const independent = point === 3, lists = [];
        for (let i = 0; i < gains; i++) {
            let cge = 1, gain = 0, cache = 1;
            if (i > 0) {
                cge = independent ? 1 : bits.read1();
                gain = cge ? this.cb.scaleFactor(bits) - 60 : 0;
                cache = Math.pow(scale, -gain);
            }
            const list = new Float32Array(META_WORDS);
            if (independent) list[0] = cache;
            else for (let b = 0; b < nb; b++) {
                if ((meta[b] >>> 12) === ZERO_BT) continue;
                if (cge === 0) {
                    let t = this.cb.scaleFactor(bits) - 60;
                    if (t !== 0) {
                        let s = 1;
                        t = gain += t;
                        if (!sign) { s -= 2 * (t & 1); t >>>= 1; }
                        cache = Math.pow(scale, -t) * s;
                    }
                }
                list[b] = cache;
            }
            lists.push(list);
        }
        return { type: 'cce', couplingPoint: independent ? 2 : point, targets: targets, gains: lists, ch: [chan], q: q, meta: meta };
    }
    /* The reference compares the coupling point with AFTER_IMDCT = 2 (cce.js:35,88,95), a value the `|=` above
     * never leaves, so it reads a per-band gain list for every coupling point; and it only steps its band index
     * on a coded band (cce.js:99-114), so it keeps testing the first ZERO band's type once it meets one.
     * quirks (default) consumes exactly those bits; otherwise the syntax of the standard is followed. */
    const quirks = this.referenceQuirks, afterImdct = !quirks && point === 3;
    for (let i = 0; i < gains; i++) {
        let cge = 1;
        if (i > 0) {
            cge = afterImdct ? 1 : bits.read1();
            if (cge) this.cb.scaleFactor(bits);
        }
        if (afterImdct) continue;
        for (let b = 0, idx = 0; b < nb; b++) {
            const coded = (this.scratchMeta[quirks ? idx : b] >>> 12) !== ZERO_BT;
            if (coded && cge === 0) this.cb.scaleFactor(bits);
            if (coded) idx++;
        }
    }
};

/* ---- raw_data_block (decoder.js:138-200) ------------------------------------------------------------ */
FrontEnd.prototype.parseRawDataBlock = function (bits, config) {
    const elements = [], parts = [];
    let hadCoupling = false;
    for (;;) {
        const type = bits.read(3);
        if (type === END) break;
        let id = bits.read(4);
        switch (type) {
        case SCE: case LFE: {
            const q = new Int16Array(FRAME), meta = new Uint16Array(META_WORDS), chan = this.ics(bits, config, null, q, meta);
            elements.push({ type: type === SCE ? 'sce' : 'lfe', id: id, commonWindow: false, maskPresent: false, hasPns: chan.hasPns, ch: [chan] });
            parts.push([q, meta]);
            break;
        }
        case CPE: {
            const q = new Int16Array(2 * FRAME), meta = new Uint16Array(2 * META_WORDS), e = this.cpe(bits, config, q, meta);
            e.id = id;
            elements.push(e); parts.push([q, meta]);
            break;
        }
        case CCE: {
            const e = this.cce(bits, config, this.keepCoupling);
            hadCoupling = true;
            if (e) { e.id = id; elements.push(e); parts.push([e.q, e.meta]); }
            break;
        }
        case DSE: {
            const align = bits.read1();
            let count = bits.read(8);
            if (count === 255) count += bits.read(8);
            if (align) bits.align();
            bits.advance(count * 8);
            break;
        }
        case PCE:
            throw new Error('TODO: PCE_ELEMENT');
        case FIL:
            if (id === 15) id += bits.read(8) - 1;
            bits.advance(id * 8);
            break;
        }
    }
    bits.align();
    let n = 0;
    for (const p of parts) n += p[0].length;
    const q = new Int16Array(n), meta = new Uint16Array(n / FRAME * META_WORDS);
    n = 0;
    for (const p of parts) { q.set(p[0], n); meta.set(p[1], n / FRAME * META_WORDS); n += p[0].length; }
    return { elements: elements, q: q, meta: meta, bitsUsed: bits.pos, hadCoupling: hadCoupling };     // bitsUsed: from the start of the frame's bytes, byte-aligned
};

/* next complete frame, or null.  `decoder.config` supplies sampleIndex (set by setCookie). */
FrontEnd.prototype.parseFrame = function (decoder) {
    const config = decoder.config;
    if (this.packets.length) {
        /* A multi-block packet: the reference keeps reading from the same bitstream after END_ELEMENT + align()
         * (decoder.js:129-199); here the rest of the packet goes back to the head of the queue.  A block that fails to
         * parse takes the rest of its packet with it. */
        const p = this.packets.shift(), bytes = p.bytes, bits = new BitStream(bytes);
        if (bits.peek(12) === 0xfff) adts.readHeader(bits);                              // decoder.js:129-130
        const frame = this.parseRawDataBlock(bits, config), used = frame.bitsUsed >>> 3;
        if (p.multi && used < bytes.length) this.packets.unshift({ bytes: bytes.subarray(used), multi: true });
        return frame;
    }
    const buf = this.buf;
    if (buf.length < 7) return null;
    const bits = new BitStream(buf), h = adts.readHeader(bits);                          // throws 'Invalid ADTS header.'
    if (h.frameLength < h.headerBytes) throw new Error('Invalid ADTS header.');
    if (buf.length < h.frameLength) return null;
    bits.end = h.frameLength * 8;
    this.buf = buf.subarray(h.frameLength);                                            // a frame that fails to parse is dropped, not retried
    return this.parseRawDataBlock(bits, config);
};

module.exports = { FrontEnd, swbOffsets, tnsCoefTable };
