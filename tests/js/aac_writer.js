/*
 * tests/js/aac_writer.js — TEST KIT: a synthetic AAC-LC bitstream writer (no AAC files exist in the container).
 *
 * Writes syntactically valid raw_data_blocks / ADTS frames (ISO/IEC 14496-3 4.4.2: SCE, CPE, LFE, CCE, DSE, FIL;
 * ics_info, section_data, scale_factor_data, pulse_data, tns_data, spectral_data) from channel descriptions, and
 * draws random descriptions.  No psychoacoustics: the spectra are random integers inside each codebook's range,
 * which is all a *parser* test needs.  Code words come from a Codebooks object (aac.js_amd/js/codebooks.js).
 *
 * A channel description:
 *   { seq, shape, groupLen[], maxSFB, globalGain, bandTypes[idx], sf[idx], q: Int16Array(1024), tns?, pulse?,
 *     split? }                        idx = g*maxSFB + sfb;  sf[idx] = the running value the standard defines for
 *                                     the band's class (spectrum / noise / intensity) after this band
 * An element: { type: 'sce'|'lfe'|'cpe'|'cce'|'dse'|'fil', id, ... }  (see writeElement)
 */
'use strict';
const path = require('path');
const { swbOffsets } = require(path.join(__dirname, '..', '..', 'aac.js_amd', 'js', 'frontend.js'));
const { UNSIGNED_BOOK } = require(path.join(__dirname, '..', '..', 'aac.js_amd', 'js', 'codebooks.js'));

function BitWriter() { this.bits = []; }
BitWriter.prototype.put = function (v, n) { for (let i = n - 1; i >= 0; i--) this.bits.push(Math.floor(v / Math.pow(2, i)) & 1); };
BitWriter.prototype.align = function () { while (this.bits.length & 7) this.bits.push(0); };
BitWriter.prototype.bytes = function () {
    this.align();
    const out = new Uint8Array(this.bits.length >> 3);
    for (let i = 0; i < this.bits.length; i++) out[i >> 3] |= this.bits[i] << (7 - (i & 7));
    return out;
};

function Rng(seed) { this.s = seed >>> 0 || 1; }
Rng.prototype.next = function () { let s = this.s; s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; this.s = s; return s; };
Rng.prototype.below = function (n) { return this.next() % n; };
Rng.prototype.chance = function (p) { return this.next() / 4294967296 < p; };

/* largest absolute value per spectral book (ISO/IEC 14496-3 table 4.A.1's LAV column); 11 escapes above 15 */
const LAV = [0, 1, 1, 2, 2, 4, 4, 7, 7, 12, 12, 8191];

function Writer(cb, sampleIndex) {
    this.cb = cb; this.sampleIndex = sampleIndex;
    this.sfCode = new Map();
    for (const e of cb.tables.sf) this.sfCode.set(e[2], e);
    this.spec = [null];
    for (let b = 1; b <= 11; b++) {
        const m = new Map();
        for (const e of cb.tables.spectral[b - 1]) m.set(e.slice(2).join(','), e);
        this.spec.push(m);
    }
}

Writer.prototype.putSf = function (w, delta) {
    const e = this.sfCode.get(delta + 60);
    if (!e) throw new Error('writer: scalefactor step out of range: ' + delta);
    w.put(e[1], e[0]);
};

Writer.prototype.putSpectral = function (w, book, vals) {
    const unsigned = UNSIGNED_BOOK[book];
    const key = vals.map(function (v) { const m = unsigned ? Math.abs(v) : v; return book === 11 && m > 16 ? 16 : m; }).join(',');
    const e = this.spec[book].get(key);
    if (!e) throw new Error('writer: book ' + book + ' has no entry for ' + key);
    w.put(e[1], e[0]);
    if (!unsigned) return;
    for (const v of vals) if (v) w.put(v < 0 ? 1 : 0, 1);
    if (book === 11)
        for (const v of vals) {
            const m = Math.abs(v);
            if (m < 16) continue;
            let n = 4;
            while (m >= (2 << n)) n++;
            for (let i = 4; i < n; i++) w.put(1, 1);
            w.put(0, 1);
            w.put(m - (1 << n), n);
        }
};

Writer.prototype.putIcsInfo = function (w, ch) {
    w.put(0, 1);
    w.put(ch.seq, 2); w.put(ch.shape, 1);
    if (ch.seq === 2) {
        w.put(ch.maxSFB, 4);
        for (let g = 0; g < ch.groupLen.length; g++)
            for (let k = 0; k < ch.groupLen[g]; k++) if (g || k) w.put(k ? 1 : 0, 1);      // 7 bits: 1 = same group as the window before
    } else {
        w.put(ch.maxSFB, 6);
        w.put(0, 1);                                         // predictor_data_present
    }
};

Writer.prototype.putIcs = function (w, ch, common) {
    const short = ch.seq === 2, groups = ch.groupLen.length, maxSFB = ch.maxSFB;
    w.put(ch.globalGain, 8);
    if (!common) this.putIcsInfo(w, ch);
    /* section_data: runs of one band type, cut at the split points the description asks for */
    const lenBits = short ? 3 : 5, esc = (1 << lenBits) - 1;
    for (let g = 0; g < groups; g++)
        for (let k = 0; k < maxSFB;) {
            let end = k + 1;
            while (end < maxSFB && ch.bandTypes[g * maxSFB + end] === ch.bandTypes[g * maxSFB + k] && !(ch.split && ch.split[g * maxSFB + end])) end++;
            w.put(ch.bandTypes[g * maxSFB + k], 4);
            let len = end - k;
            while (len >= esc) { w.put(esc, lenBits); len -= esc; }
            w.put(len, lenBits);
            k = end;
        }
    /* scale_factor_data */
    let spec = ch.globalGain, noise = ch.globalGain - 90, inten = 0, first = true;
    for (let idx = 0; idx < groups * maxSFB; idx++) {
        const bt = ch.bandTypes[idx];
        if (bt === 0) continue;
        if (bt >= 14) { this.putSf(w, ch.sf[idx] - inten); inten = ch.sf[idx]; }
        else if (bt === 13) {
            if (first) { w.put(ch.sf[idx] - noise + 256, 9); first = false; } else this.putSf(w, ch.sf[idx] - noise);
            noise = ch.sf[idx];
        } else { this.putSf(w, ch.sf[idx] - spec); spec = ch.sf[idx]; }
    }
    /* pulse_data */
    if (ch.pulse) {
        w.put(1, 1);
        w.put(ch.pulse.offset.length - 1, 2); w.put(ch.pulse.swb, 6);
        let at = swbOffsets(this.sampleIndex, false)[ch.pulse.swb];
        for (let i = 0; i < ch.pulse.offset.length; i++) { w.put(ch.pulse.offset[i] - at, 5); at = ch.pulse.offset[i]; w.put(ch.pulse.amp[i], 4); }
    } else w.put(0, 1);
    /* tns_data: ch.tns = { res[w], filt[w] = [{length, order, direction, compress, field[]}] } */
    if (ch.tns) {
        w.put(1, 1);
        for (let win = 0; win < (short ? 8 : 1); win++) {
            const f = ch.tns.filt[win];
            w.put(f.length, short ? 1 : 2);
            if (!f.length) continue;
            w.put(ch.tns.res[win], 1);
            for (const x of f) {
                w.put(x.length, short ? 4 : 6); w.put(x.order, short ? 3 : 5);
                if (!x.order) continue;
                w.put(x.direction ? 1 : 0, 1); w.put(x.compress, 1);
                for (const c of x.field) w.put(c, ch.tns.res[win] + 3 - x.compress);
            }
        }
    } else w.put(0, 1);
    w.put(ch.gainControl ? 1 : 0, 1);
    /* spectral_data */
    const off = swbOffsets(this.sampleIndex, short);
    for (let g = 0, groupOff = 0; g < groups; g++) {
        for (let sfb = 0; sfb < maxSFB; sfb++) {
            const bt = ch.bandTypes[g * maxSFB + sfb];
            if (bt === 0 || bt >= 12) continue;                  // 12 is reserved: only written by the malformed-input tests
            const step = bt < 5 ? 4 : 2;
            for (let win = 0; win < ch.groupLen[g]; win++)
                for (let k = off[sfb]; k < off[sfb + 1]; k += step)
                    this.putSpectral(w, bt, Array.prototype.slice.call(ch.q, groupOff + win * 128 + k, groupOff + win * 128 + k + step));
        }
        groupOff += ch.groupLen[g] * 128;
    }
};

const TYPE_CODE = { sce: 0, cpe: 1, cce: 2, lfe: 3, dse: 4, pce: 5, fil: 6 };

Writer.prototype.putElement = function (w, e) {
    w.put(TYPE_CODE[e.type], 3);
    switch (e.type) {
    case 'sce': case 'lfe':
        w.put(e.id, 4); this.putIcs(w, e.ch[0], false);
        break;
    case 'cpe':
        w.put(e.id, 4);
        w.put(e.commonWindow ? 1 : 0, 1);
        if (e.commonWindow) {
            this.putIcsInfo(w, e.ch[0]);
            w.put(e.mask, 2);                                // 0 none, 1 per band, 2 all
            if (e.mask === 1) for (let i = 0; i < e.ch[0].groupLen.length * e.ch[0].maxSFB; i++) w.put(e.msUsed[i], 1);
        }
        this.putIcs(w, e.ch[0], e.commonWindow); this.putIcs(w, e.ch[1], e.commonWindow);
        break;
    case 'cce': {
        /* e: { id, point (0..3 as coded: ind_sw_cce_flag<<1 | cc_domain), targets: [{pair, id, sel}], sign, scale, ch,
         *      lists: [{cge, common, steps[]}] for gain element lists 1.., quirks } */
        w.put(e.id, 4);
        w.put(e.point >> 1, 1); w.put(e.targets.length - 1, 3);
        for (const t of e.targets) { w.put(t.pair ? 1 : 0, 1); w.put(t.id, 4); if (t.pair) w.put(t.sel, 2); }
        w.put(e.point & 1, 1); w.put(e.sign, 1); w.put(e.scale, 2);
        this.putIcs(w, e.ch[0], false);
        const coupling = (e.point & 2) | (e.point & 1) | ((e.point >> 1) & 1), after = !e.quirks && coupling === 3;
        const nb = e.ch[0].groupLen.length * e.ch[0].maxSFB;
        for (const l of e.lists) {
            if (!after) w.put(l.cge, 1);
            if (after || l.cge) this.putSf(w, l.common);
            if (after) continue;
            let s = 0;
            for (let b = 0, idx = 0; b < nb; b++) {
                const coded = e.ch[0].bandTypes[e.quirks ? idx : b] !== 0;
                if (coded && !l.cge) this.putSf(w, l.steps[s++ % l.steps.length]);
                if (coded) idx++;
            }
        }
        break;
    }
    case 'dse':
        w.put(e.id, 4);
        w.put(e.align ? 1 : 0, 1);
        if (e.bytes.length >= 255) { w.put(255, 8); w.put(e.bytes.length - 255, 8); } else w.put(e.bytes.length, 8);
        if (e.align) w.align();
        for (const b of e.bytes) w.put(b, 8);
        break;
    case 'pce':
        w.put(e.id, 4);
        break;
    case 'fil':
        if (e.count >= 15) { w.put(15, 4); w.put(e.count - 14, 8); } else w.put(e.count, 4);
        for (let i = 0; i < e.count; i++) w.put(0xA5, 8);
        break;
    }
};

Writer.prototype.rawDataBlock = function (elements) {
    const w = new BitWriter();
    for (const e of elements) this.putElement(w, e);
    w.put(7, 3);
    return w.bytes();
};

/* ADTS frame around a raw_data_block; opts.crc adds the 16-bit CRC field (content not checked by either decoder) */
Writer.prototype.adtsFrame = function (elements, chanConfig, opts) {
    opts = opts || {};
    const body = this.rawDataBlock(elements), w = new BitWriter(), header = opts.crc ? 9 : 7;
    w.put(0xfff, 12); w.put(0, 1); w.put(0, 2); w.put(opts.crc ? 0 : 1, 1);
    w.put(1, 2); w.put(this.sampleIndex, 4); w.put(0, 1); w.put(chanConfig, 3); w.put(0, 4);
    w.put(header + body.length, 13); w.put(0x7ff, 11); w.put(0, 2);
    if (opts.crc) w.put(0xBEEF, 16);
    const head = w.bytes(), out = new Uint8Array(head.length + body.length);
    out.set(head); out.set(body, head.length);
    return out;
};

/* ---- random descriptions -------------------------------------------------------------------------------- */
const GROUPINGS = [[8], [3, 4, 1], [1, 1, 1, 1, 1, 1, 1, 1], [2, 6], [4, 4], [1, 7]];

/* opts: seq, shape, allowIS, allowPNS, tns, pulse, template (share ics_info with this channel), maxSFB */
Writer.prototype.randomChannel = function (rng, opts) {
    opts = opts || {};
    const t = opts.template, seq = t ? t.seq : (opts.seq === undefined ? rng.below(4) : opts.seq), short = seq === 2;
    const off = swbOffsets(this.sampleIndex, short), swbCount = off.length - 1;
    const ch = { seq: seq, shape: t ? t.shape : (opts.shape === undefined ? rng.below(2) : opts.shape),
                 groupLen: t ? t.groupLen : (short ? GROUPINGS[rng.below(GROUPINGS.length)] : [1]),
                 maxSFB: t ? t.maxSFB : (opts.maxSFB !== undefined ? opts.maxSFB : (short ? Math.min(swbCount, 4 + rng.below(swbCount - 3)) : Math.min(swbCount, 20 + rng.below(swbCount - 19)))),
                 globalGain: 100 + rng.below(20), q: new Int16Array(1024) };     // table index 200..220: PCM near -15 dBFS
    const nb = ch.groupLen.length * ch.maxSFB;
    ch.bandTypes = new Uint8Array(nb); ch.sf = new Int32Array(nb); ch.split = new Uint8Array(nb);
    let spec = ch.globalGain, noise = ch.globalGain - 90, inten = 0, first = true;
    for (let g = 0; g < ch.groupLen.length; g++)
        for (let k = 0; k < ch.maxSFB;) {
            const run = Math.min(ch.maxSFB - k, rng.chance(0.1) ? 31 + rng.below(8) : 1 + rng.below(7)), r = rng.below(100);
            let bt;
            if (r < 10) bt = 0;
            else if (opts.allowIS && r < 25) bt = 14 + rng.below(2);
            else if (opts.allowPNS && r < 33) bt = 13;
            else if (r < 55) bt = 1 + rng.below(4);
            else if (r < 88) bt = 5 + rng.below(6);
            else bt = 11;
            ch.split[g * ch.maxSFB + k] = 1;
            for (let i = 0; i < run; i++, k++) {
                const idx = g * ch.maxSFB + k;
                ch.bandTypes[idx] = bt;
                if (rng.chance(0.05)) ch.split[idx] = 1;
                if (bt === 0) continue;
                if (bt >= 14) { inten = Math.max(-40, Math.min(40, inten + rng.below(21) - 10)); ch.sf[idx] = inten; }
                else if (bt === 13) {
                    noise = first ? noise + rng.below(64) - 16 : Math.max(-60, Math.min(120, noise + rng.below(13) - 6));
                    first = false; ch.sf[idx] = noise;
                } else { spec = Math.max(85, Math.min(125, spec + rng.below(17) - 8)); ch.sf[idx] = spec; }
            }
        }
    for (let g = 0, groupOff = 0; g < ch.groupLen.length; g++) {
        for (let sfb = 0; sfb < ch.maxSFB; sfb++) {
            const bt = ch.bandTypes[g * ch.maxSFB + sfb];
            if (bt === 0 || bt >= 13) continue;
            const signed = !UNSIGNED_BOOK[bt], lav = Math.min(LAV[bt], 15);
            for (let win = 0; win < ch.groupLen[g]; win++)
                for (let k = off[sfb]; k < off[sfb + 1]; k++) {
                    let v = rng.below(2 * lav + 1) - lav;
                    if (rng.chance(0.35)) v = 0;
                    if (bt === 11 && rng.chance(0.06)) v = (rng.below(2) ? 1 : -1) * (16 + rng.below(rng.chance(0.2) ? 8175 : 200));
                    ch.q[groupOff + win * 128 + k] = v;
                }
        }
        groupOff += ch.groupLen[g] * 128;
    }
    if (opts.pulse && !short) {
        const n = 1 + rng.below(4), swb = rng.below(Math.min(swbCount, 30));
        ch.pulse = { swb: swb, offset: [], amp: [] };
        for (let i = 0, at = off[swb]; i < n; i++) { at += rng.below(32); ch.pulse.offset.push(at); ch.pulse.amp.push(rng.below(16)); }
    }
    if (opts.tns) {
        ch.tns = { res: [], filt: [] };
        for (let win = 0; win < (short ? 8 : 1); win++) {
            const nf = short ? rng.below(2) : rng.below(4), res = rng.below(2), list = [];
            for (let f = 0; f < nf; f++) {
                const order = rng.chance(0.15) ? 0 : 1 + rng.below(short ? 7 : 12), compress = rng.below(2), field = [];
                for (let i = 0; i < order; i++) field.push(rng.below(1 << (res + 3 - compress)));
                list.push({ length: rng.below(short ? 14 : 40), order: order, direction: rng.below(2) === 1, compress: compress, field: field });
            }
            ch.tns.res.push(res); ch.tns.filt.push(list);
        }
    }
    return ch;
};

module.exports = { Writer, BitWriter, Rng, LAV };
