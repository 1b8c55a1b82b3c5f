#!/usr/bin/env node
/*
 * gen_golden.js — produce tests/golden/golden.{bin,json} by RUNNING the reference
 * (audiocogs/aac.js, read-only at /root/reference) under Node.
 *
 * Runs only in the build container: it refuses to start when /root/reference is
 * absent (the GPU box).  Nothing from the reference is copied; only numbers it computes
 * (inputs chosen here + the outputs it returns) are stored.
 *
 *   NODE_PATH=tests/golden/gen/stubs node tests/golden/gen/gen_golden.js
 *
 * Vectors (SURVEY.md §8c F1..F7):
 *   tables.*     IQ / SF / windows / FFT roots / MDCT twiddles / SWB offsets
 *   fft*, imdct* seeded inputs and outputs of FFT.process / MDCT.process
 *   pns.*        the PNS generator sequence exactly as written (known-bad, ics.js:234)
 *   scn_*        multi-frame scenarios through the reference's own readChunk():
 *                process(elements) + interleave, with dequant through the real
 *                ICStream.decodeSpectralData (Huffman stubbed with a feeder)
 */
'use strict';
const fs = require('fs');
const path = require('path');

const REF = '/root/reference/src/';
if (!fs.existsSync(REF + 'decoder.js')) {
    console.error('gen_golden.js: /root/reference not present - golden vectors can only be generated in the build container');
    process.exit(2);
}
process.env.NODE_PATH = path.join(__dirname, 'stubs');
require('module').Module._initPaths();

const tables = require(REF + 'tables.js');
const FFT = require(REF + 'fft.js');
const MDCT = require(REF + 'mdct.js');
const mdctTables = require(REF + 'mdct_tables.js');
const FilterBank = require(REF + 'filter_bank.js');
const ICStream = require(REF + 'ics.js');
const CPEElement = require(REF + 'cpe.js');
const Huffman = require(REF + 'huffman.js');
const AACDecoder = require(REF + 'decoder.js');

// ---------------------------------------------------------------- output container
const chunks = [];
const manifest = {};
let cursor = 0;
function put(name, typed, shape) {
    const buf = Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength);
    const dtype = typed instanceof Float32Array ? 'f4' : typed instanceof Float64Array ? 'f8' :
                  typed instanceof Int32Array ? 'i4' : typed instanceof Int16Array ? 'i2' :
                  typed instanceof Uint16Array ? 'u2' : typed instanceof Uint8Array ? 'u1' : null;
    if (!dtype) throw new Error('dtype? ' + name);
    manifest[name] = { offset: cursor, dtype: dtype, shape: shape || [typed.length] };
    chunks.push(Buffer.from(buf));      // copy: the source may be reused
    cursor += buf.length;
    const pad = (8 - (cursor % 8)) % 8;
    if (pad) { chunks.push(Buffer.alloc(pad)); cursor += pad; }
}

// ---------------------------------------------------------------- PRNG (documented: xorshift32)
function Rng(seed) { this.s = seed >>> 0; }
Rng.prototype.next = function () {
    let s = this.s;
    s ^= s << 13; s >>>= 0;
    s ^= s >>> 17;
    s ^= s << 5;  s >>>= 0;
    this.s = s;
    return s;
};
Rng.prototype.unit = function () { return (this.next() | 0) / 2147483648; };     // [-1, 1)
Rng.prototype.below = function (n) { return this.next() % n; };

// ---------------------------------------------------------------- F1: tables
put('tables.iq', tables.IQ_TABLE);
put('tables.sf', tables.SCALEFACTOR_TABLE);

// windows are module-private (filter_bank.js:81-86): recover them by making the IMDCT return 1.0
function recoverWindows() {
    const fb = new FilterBank(false, 1);
    const ones = function (input, inOff, buf, outOff) { for (let i = 0; i < 2048; i++) buf[i] = 1; };
    fb.mdctLong.process = ones;
    const zero = new Float32Array(1024);
    const out = {};
    for (let shape = 0; shape < 2; shape++) {
        const o = new Float32Array(1024);
        fb.overlaps[0].fill(0);
        fb.process({ windowSequence: 0, windowShape: [shape, shape] }, zero, o, 0);
        const long = Float32Array.from(o);                         // out[i] = Wprev_long[i]
        const rev = fb.overlaps[0];                                  // ov[i]  = W_long[1023-i]
        for (let i = 0; i < 1024; i++) if (rev[1023 - i] !== long[i]) throw new Error('window recovery mismatch');
        fb.overlaps[0].fill(0);
        fb.process({ windowSequence: 3, windowShape: [shape, shape] }, zero, o, 0);
        const short = Float32Array.from(o.subarray(448, 576));       // out[448+i] = Wprev_short[i]
        fb.overlaps[0].fill(0);
        fb.process({ windowSequence: 1, windowShape: [shape, shape] }, zero, o, 0);
        for (let i = 0; i < 128; i++) if (fb.overlaps[0][448 + i] !== short[127 - i]) throw new Error('short window recovery mismatch');
        out[shape] = { long: long, short: short };
    }
    return out;
}
const W = recoverWindows();
put('tables.sine_long', W[0].long);  put('tables.kbd_long', W[1].long);
put('tables.sine_short', W[0].short); put('tables.kbd_short', W[1].short);

function flatten(rows, width, Type) {
    const a = new Type(rows.length * width);
    for (let i = 0; i < rows.length; i++) for (let j = 0; j < width; j++) a[i * width + j] = rows[i][j];
    return a;
}
put('tables.fft_roots_512', flatten(new FFT(512).roots, 3, Float32Array), [512, 3]);
put('tables.fft_roots_64', flatten(new FFT(64).roots, 2, Float32Array), [64, 2]);
put('tables.mdct_2048', flatten(mdctTables.MDCT_TABLE_2048, 2, Float64Array), [512, 2]);
put('tables.mdct_256', flatten(mdctTables.MDCT_TABLE_256, 2, Float64Array), [64, 2]);
{
    // SWB offsets, all 12 sample-rate indices, padded rows; counts alongside (tables.js:34-163)
    const long = new Uint16Array(12 * 64), short = new Uint16Array(12 * 16), counts = new Int32Array(24);
    for (let s = 0; s < 12; s++) {
        const l = tables.SWB_OFFSET_1024[s], sh = tables.SWB_OFFSET_128[s];
        for (let i = 0; i < l.length; i++) long[s * 64 + i] = l[i];
        for (let i = 0; i < sh.length; i++) short[s * 16 + i] = sh[i];
        counts[s] = tables.SWB_LONG_WINDOW_COUNT[s];
        counts[12 + s] = tables.SWB_SHORT_WINDOW_COUNT[s];
        if (l.length - 1 !== counts[s] || sh.length - 1 !== counts[12 + s]) throw new Error('swb count mismatch ' + s);
    }
    put('tables.swb_long', long, [12, 64]);
    put('tables.swb_short', short, [12, 16]);
    put('tables.swb_counts', counts, [2, 12]);
}

// ---------------------------------------------------------------- F2: FFT and IMDCT
for (const len of [512, 64]) {
    const fft = new FFT(len);
    const nvec = 3, inp = new Float32Array(nvec * len * 2), outp = new Float32Array(nvec * len * 2);
    for (let v = 0; v < nvec; v++) {
        const rng = new Rng(0xAAC0F000 + len + v);
        const buf = [];
        for (let i = 0; i < len; i++) {
            buf.push(new Float32Array([rng.unit() * 1000, rng.unit() * 1000]));
            inp[(v * len + i) * 2] = buf[i][0]; inp[(v * len + i) * 2 + 1] = buf[i][1];
        }
        fft.process(buf, false);
        for (let i = 0; i < len; i++) { outp[(v * len + i) * 2] = buf[i][0]; outp[(v * len + i) * 2 + 1] = buf[i][1]; }
    }
    put('fft' + len + '.in', inp, [nvec, len, 2]);
    put('fft' + len + '.out', outp, [nvec, len, 2]);
}
for (const N of [2048, 256]) {
    const mdct = new MDCT(N);
    const nvec = 4, inp = new Float32Array(nvec * N / 2), outp = new Float32Array(nvec * N);
    for (let v = 0; v < nvec; v++) {
        const rng = new Rng(0xAAC0D000 + N + v);
        const x = new Float32Array(N / 2), y = new Float32Array(N);
        for (let k = 0; k < N / 2; k++) x[k] = 4000 * rng.unit() * Math.exp(-k / (N / 8));
        mdct.process(x, 0, y, 0);
        inp.set(x, v * N / 2); outp.set(y, v * N);
    }
    put('imdct' + N + '.in', inp, [nvec, N / 2]);
    put('imdct' + N + '.out', outp, [nvec, N]);
}

// ---------------------------------------------------------------- F5: PNS generator as written
{
    const seq = new Int32Array(16);
    let randomState = 0x1F2E3D4C;                                   // ics.js:31
    for (let i = 0; i < 16; i++) {
        randomState = (randomState * (1664525 + 1013904223)) | 0;  // the expression of ics.js:234
        seq[i] = randomState;
    }
    // cross-check against the real code path: one NOISE band of width 16
    const ics = new ICStream({ frameLength: 1024, sampleIndex: 3 });
    ics.info.groupCount = 1; ics.info.groupLength[0] = 1; ics.info.maxSFB = 1;
    ics.info.swbOffsets = new Uint16Array([0, 16]);
    ics.bandTypes[0] = 13; ics.scaleFactors[0] = 1;
    const probe = [];
    const origF32 = ics.data;
    ics.data = new Proxy(origF32, { set: function (t, p, v) { if (probe.length < 16 && Number.isInteger(v)) probe.push(v); t[p] = v; return true; } });
    ics.decodeSpectralData(null);
    for (let i = 0; i < 16; i++) if (probe[i] !== seq[i]) throw new Error('PNS probe mismatch at ' + i);
    put('pns.sequence', seq);
}

// ---------------------------------------------------------------- scenarios through readChunk()
const SAMPLE_INDEX = 3;                                             // 48 kHz
function makeDecoder(channels) {
    const dec = new AACDecoder();
    dec.format = {};
    dec.config = { profile: 2, sampleIndex: SAMPLE_INDEX, sampleRate: 48000, chanConfig: channels, frameLength: 1024 };
    dec.filter_bank = new FilterBank(false, channels);              // decoder.js:112
    dec.bitstream = { peek: function () { return 0; }, read: function () { return 7; }, align: function () {}, advance: function () {} };
    return dec;
}

// feeder for the spectral Huffman decode: the real dequant loop (ics.js:244-256) pulls from here
let feed = null, feedPos = 0;
Huffman.decodeSpectralData = function (stream, cb, data, off) {
    const num = cb >= 5 ? 2 : 4;
    for (let j = 0; j < num; j++) data[off + j] = feed[feedPos++];
};

const GROUPINGS = [[8], [3, 4, 1], [1, 1, 1, 1, 1, 1, 1, 1], [2, 6], [4, 4], [1, 7]];

// random side info + quantised spectrum for one channel
function randomChannel(rng, seq, shape, opts) {
    const short = seq === 2;
    const groupLen = opts.groupLen ? opts.groupLen : (short ? GROUPINGS[rng.below(GROUPINGS.length)] : [1]);
    const swbCount = short ? 14 : 49;
    const maxSFB = opts.maxSFB !== undefined ? opts.maxSFB : (short ? 6 + rng.below(9) : 25 + rng.below(25));
    const nb = groupLen.length * maxSFB;
    const bandTypes = new Int32Array(nb), sfIdx = new Int32Array(nb);
    for (let g = 0; g < groupLen.length; g++) {
        let k = 0;
        while (k < maxSFB) {
            const run = Math.min(maxSFB - k, 1 + rng.below(6));
            const r = rng.below(100);
            let bt;
            if (r < 12) bt = 0;
            else if (opts.allowIS && r < 30) bt = 14 + rng.below(2);
            else if (r < 55) bt = 1 + rng.below(4);
            else if (r < 90) bt = 5 + rng.below(6);
            else bt = 11;
            for (let i = 0; i < run; i++, k++) {
                bandTypes[g * maxSFB + k] = bt;
                // normal bands: offset[0]-100+200 (ics.js:166); intensity: 200-clamp (ics.js:141-145)
                sfIdx[g * maxSFB + k] = (bt === 14 || bt === 15) ? 188 + rng.below(24) : (bt === 0 ? 0 : 200 + rng.below(30));   // PCM lands near -15 dBFS
            }
        }
    }
    const q = new Int16Array(1024);
    const offsets = short ? tables.SWB_OFFSET_128[SAMPLE_INDEX] : tables.SWB_OFFSET_1024[SAMPLE_INDEX];
    if (opts.garbage) for (let i = 0; i < 1024; i++) q[i] = (rng.below(41) - 20);    // must be ignored outside coded bands
    let groupOff = 0;
    for (let g = 0; g < groupLen.length; g++) {
        for (let sfb = 0; sfb < maxSFB; sfb++) {
            const bt = bandTypes[g * maxSFB + sfb];
            const coded = !(bt === 0 || bt === 14 || bt === 15);
            for (let w = 0; w < groupLen[g]; w++) {
                for (let k = offsets[sfb]; k < offsets[sfb + 1]; k++) {
                    if (!coded) continue;
                    const kk = short ? k * 8 : k;
                    const amp = Math.max(1, Math.floor(24 * Math.exp(-kk / 180)));
                    let v = rng.below(2 * amp + 1) - amp;
                    if (bt === 11 && rng.below(48) === 0) v = (rng.below(2) ? 1 : -1) * (17 + rng.below(8174));   // escape range, <= 8190
                    q[groupOff + w * 128 + k] = v;
                }
            }
        }
        groupOff += groupLen[g] * 128;
    }
    return { seq: seq, shape: shape, short: short, groupLen: groupLen, maxSFB: maxSFB, swbCount: swbCount,
             bandTypes: bandTypes, sfIdx: sfIdx, q: q };
}

// build a reference ICStream from a channel description and run the real dequant on it
function buildICS(dec, ch, sharedInfo) {
    const ics = new ICStream(dec.config);
    if (sharedInfo) ics.info = sharedInfo;                           // cpe.js:44
    const info = ics.info;
    if (!sharedInfo) {
        info.windowSequence = ch.seq;
        info.windowShape[1] = ch.shape;                              // [0] stays 0: fresh ICSInfo per frame (decoder.js:145,153)
        info.groupCount = ch.groupLen.length;
        for (let g = 0; g < ch.groupLen.length; g++) info.groupLength[g] = ch.groupLen[g];
        info.maxSFB = ch.maxSFB;
        info.windowCount = ch.short ? 8 : 1;
        info.swbOffsets = ch.short ? tables.SWB_OFFSET_128[SAMPLE_INDEX] : tables.SWB_OFFSET_1024[SAMPLE_INDEX];
        info.swbCount = ch.swbCount;
    }
    const nb = ch.groupLen.length * ch.maxSFB;
    for (let g = 0; g < ch.groupLen.length; g++) {
        for (let k = 0; k < ch.maxSFB;) {                            // sectEnd: runs of one band type (ics.js:83-116)
            let end = k;
            while (end < ch.maxSFB && ch.bandTypes[g * ch.maxSFB + end] === ch.bandTypes[g * ch.maxSFB + k]) end++;
            for (; k < end; k++) { ics.bandTypes[g * ch.maxSFB + k] = ch.bandTypes[g * ch.maxSFB + k]; ics.sectEnd[g * ch.maxSFB + k] = end; }
        }
    }
    for (let i = 0; i < nb; i++)
        ics.scaleFactors[i] = ch.bandTypes[i] === 0 ? 0 : tables.SCALEFACTOR_TABLE[ch.sfIdx[i]];
    // feeder order = iteration order of ics.js:212-258
    const order = [];
    const offsets = info.swbOffsets;
    let groupOff = 0;
    for (let g = 0; g < ch.groupLen.length; g++) {
        for (let sfb = 0; sfb < ch.maxSFB; sfb++) {
            const bt = ch.bandTypes[g * ch.maxSFB + sfb];
            if (!(bt === 0 || bt === 14 || bt === 15))
                for (let w = 0; w < ch.groupLen[g]; w++)
                    for (let k = offsets[sfb]; k < offsets[sfb + 1]; k++) order.push(ch.q[groupOff + w * 128 + k]);
        }
        groupOff += ch.groupLen[g] * 128;
    }
    feed = order; feedPos = 0;
    ics.decodeSpectralData(null);
    if (feedPos !== order.length) throw new Error('feeder not drained');
    if (ch.tns) {                                                    // TNS side info present: tns.process must still be a no-op
        ics.tnsPresent = 1;
        const n = ch.short ? 8 : 1;
        for (let w = 0; w < n; w++) {
            ics.tns.nFilt[w] = 1; ics.tns.length[w][0] = ch.short ? 10 : 20; ics.tns.order[w][0] = ch.short ? 7 : 12;
            ics.tns.direction[w][0] = !!(w & 1);
            for (let i = 0; i < 12; i++) ics.tns.coef[w][0][i] = 0.3 - 0.05 * i;
        }
    }
    return ics;
}

function packMeta(ch, msUsed) {
    const m = new Uint16Array(120);
    const nb = ch.groupLen.length * ch.maxSFB;
    for (let i = 0; i < nb; i++)
        m[i] = (ch.sfIdx[i] & 0x1ff) | ((msUsed && msUsed[i]) ? 0x400 : 0) | (ch.bandTypes[i] << 12);
    return m;
}
function packChanInfo(view, off, ch) {
    view.setUint8(off + 0, ch.seq); view.setUint8(off + 1, ch.shape); view.setUint8(off + 2, 0);
    view.setUint8(off + 3, ch.maxSFB); view.setUint8(off + 4, ch.groupLen.length);
    for (let g = 0; g < ch.groupLen.length; g++) view.setUint8(off + 8 + g, ch.groupLen[g]);
}

/*
 * layout: array of element kinds per frame, e.g. ['cpe'] or ['cpe','cpe','cpe','sce'].
 * Every frame goes through dec.readChunk() with process() fed the hand-built elements.
 */
function scenario(name, seed, nFrames, layout, seqPattern, opts) {
    opts = opts || {};
    const C = layout.reduce(function (a, e) { return a + (e === 'cpe' ? 2 : 1); }, 0);
    const dec = makeDecoder(C);
    const rng = new Rng(seed);
    const unitsPerFrame = layout.length;
    const unitBuf = new ArrayBuffer(64 * unitsPerFrame * nFrames), view = new DataView(unitBuf);
    const qAll = new Int16Array(nFrames * C * 1024), metaAll = new Uint16Array(nFrames * C * 120);
    const specAll = new Float32Array(nFrames * C * 1024), pcmAll = new Float32Array(nFrames * C * 1024);
    let block = 0, unitIdx = 0;
    for (let t = 0; t < nFrames; t++) {
        const elements = [], chanDescs = [];
        let channel = 0;
        for (let e = 0; e < layout.length; e++) {
            const uoff = 64 * unitIdx;
            view.setUint32(uoff + 0, opts.stream || 0, true);
            view.setUint32(uoff + 4, t * C * 1024, true);
            view.setUint16(uoff + 8, channel, true);
            view.setUint16(uoff + 10, C, true);
            view.setUint32(uoff + 16, block, true);
            view.setUint32(uoff + 20, block, true);
            if (layout[e] === 'cpe') {
                const common = opts.noCommon ? false : (rng.below(8) !== 0);
                const seqL = seqPattern[(t + e) % seqPattern.length];
                const seqR = common ? seqL : seqPattern[(t + e + 3) % seqPattern.length];
                const L = randomChannel(rng, seqL, rng.below(2), { garbage: t % 3 === 1 });
                const R = randomChannel(rng, seqR, common ? L.shape : rng.below(2),
                                        { allowIS: true, garbage: t % 3 === 1,
                                          // one shared ICSInfo (cpe.js:44) => same grouping and maxSFB
                                          maxSFB: common ? L.maxSFB : undefined, groupLen: common ? L.groupLen : undefined });
                L.tns = R.tns = !!opts.tns && (t % 2 === 0);
                const maskMode = rng.below(4);           // 0: none, 1/2: per-band, 3: all ones (cpe.js:49-70)
                const nb = L.groupLen.length * L.maxSFB;
                const cpe = new CPEElement(dec.config);
                cpe.commonWindow = common;
                cpe.maskPresent = common && maskMode !== 0;
                const msUsed = new Array(128).fill(false);
                if (common && maskMode === 3) msUsed.fill(true);
                else if (common && maskMode !== 0) for (let i = 0; i < nb; i++) msUsed[i] = rng.below(2) === 1;
                cpe.ms_used = msUsed;
                cpe.left = buildICS(dec, L, null);
                cpe.right = buildICS(dec, R, common ? cpe.left.info : null);
                elements.push(cpe);
                chanDescs.push({ elem: cpe, L: L, R: R });
                view.setUint8(uoff + 12, 2);
                view.setUint8(uoff + 13, (common ? 1 : 0) | (cpe.maskPresent ? 2 : 0));
                packChanInfo(view, uoff + 24, L); packChanInfo(view, uoff + 40, R);
                qAll.set(L.q, block * 1024); qAll.set(R.q, (block + 1) * 1024);
                metaAll.set(packMeta(L, msUsed), block * 120); metaAll.set(packMeta(R, null), (block + 1) * 120);
                block += 2; channel += 2;
            } else {
                const S = randomChannel(rng, seqPattern[(t + e) % seqPattern.length], rng.below(2), { garbage: t % 3 === 2 });
                S.tns = !!opts.tns && (t % 2 === 1);
                const ics = buildICS(dec, S, null);
                elements.push(ics);
                chanDescs.push({ elem: ics, S: S });
                view.setUint8(uoff + 12, 1);
                packChanInfo(view, uoff + 24, S);
                qAll.set(S.q, block * 1024);
                metaAll.set(packMeta(S, null), block * 120);
                block += 1; channel += 1;
            }
            unitIdx++;
        }
        dec.process = function () { return AACDecoder.prototype.process.call(this, elements); };
        const pcm = dec.readChunk();                                 // decoder.js:125-216
        if (pcm.length !== 1024 * C) throw new Error('unexpected readChunk size');
        pcmAll.set(pcm, t * C * 1024);
        // spectra as they entered the filterbank (after MS / IS, decoder.js:294-302)
        let b = block - C;
        for (const d of chanDescs) {
            if (d.L) { specAll.set(d.elem.left.data, b * 1024); specAll.set(d.elem.right.data, (b + 1) * 1024); b += 2; }
            else { specAll.set(d.elem.data, b * 1024); b += 1; }
        }
    }
    put(name + '.units', new Uint8Array(unitBuf), [unitsPerFrame * nFrames, 64]);
    put(name + '.q', qAll, [nFrames * C, 1024]);
    put(name + '.meta', metaAll, [nFrames * C, 120]);
    put(name + '.spec', specAll, [nFrames * C, 1024]);
    put(name + '.pcm', pcmAll, [nFrames, 1024, C]);
    const ov = new Float32Array(C * 1024);
    for (let c = 0; c < C; c++) ov.set(dec.filter_bank.overlaps[c], c * 1024);
    put(name + '.overlap', ov, [C, 1024]);
}

// F7 / config 1: one mono ONLY_LONG sine frame, spectrum 1200000*u*exp(-k/200) (SURVEY.md §8d)
{
    const dec = makeDecoder(1);
    const rng = new Rng(0xAAC00001);
    const ics = new ICStream(dec.config);
    ics.info.windowSequence = 0; ics.info.windowShape[1] = 0; ics.info.groupCount = 1; ics.info.groupLength[0] = 1;
    ics.info.maxSFB = 49; ics.info.windowCount = 1; ics.info.swbOffsets = tables.SWB_OFFSET_1024[SAMPLE_INDEX]; ics.info.swbCount = 49;
    for (let k = 0; k < 1024; k++) ics.data[k] = 1200000 * rng.unit() * Math.exp(-k / 200);
    const spec = Float32Array.from(ics.data);
    dec.process = function () { return AACDecoder.prototype.process.call(this, [ics]); };
    const pcm = dec.readChunk();
    put('cfg1.spec', spec, [1, 1024]);
    put('cfg1.pcm', pcm, [1, 1024, 1]);
    put('cfg1.overlap', dec.filter_bank.overlaps[0], [1, 1024]);
}

// all window-sequence transitions, both shapes, MS, IS, zero bands, grouping, TNS side info present
scenario('scn_stereo', 0xAAC0B001, 18, ['cpe'], [0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 0, 0, 1, 3, 2, 0, 3, 1], { tns: true });
// L and R with different window sequences (no common window)
scenario('scn_split', 0xAAC0B002, 8, ['cpe'], [0, 1, 2, 3, 2, 0, 3, 1], { noCommon: true });
// config 5 shape: 3 CPE + LFE, chanConfig = 7 taken literally (decoder.js:219)
scenario('scn_7ch', 0xAAC0B003, 5, ['cpe', 'cpe', 'cpe', 'sce'], [0, 1, 2, 3, 0], { tns: true });
// mono SCE stream
scenario('scn_mono', 0xAAC0B004, 9, ['sce'], [0, 1, 2, 2, 3, 0, 2, 0, 0], {});

// ---------------------------------------------------------------- write
const outDir = path.join(__dirname, '..');
fs.writeFileSync(path.join(outDir, 'golden.bin'), Buffer.concat(chunks));
fs.writeFileSync(path.join(outDir, 'golden.json'), JSON.stringify({
    generator: 'tests/golden/gen/gen_golden.js',
    reference: 'audiocogs/aac.js v0.1.3 (package.json:3), node ' + process.version,
    prng: 'xorshift32 (13,17,5); unit() = int32(state) / 2^31',
    arrays: manifest
}, null, 1));
console.log('golden.bin: ' + cursor + ' bytes, ' + Object.keys(manifest).length + ' arrays');
