#!/usr/bin/env node
/*
 * tests/js/corpus_cases.js — TEST KIT: the front-end corpus (VERDICT round 5, item 5: "front-end parity rests on five short
 * streams").  A deterministic list of a few hundred short AAC-LC streams written by aac_writer.js — every one of the twelve
 * sample-rate indices the tables hold, mono / stereo / 3.0 / 5.1 / seven-channel layouts, DSE / FIL / CCE elements in between,
 * every spectral codebook including escapes up to 8190, grouped short windows, split and common windows with all three M/S
 * modes, intensity bands, TNS and pulse side info — plus malformed variants (one bad frame behind good ones, one per error the
 * reference raises: src/ics.js:65-198, src/decoder.js:183-196).
 *
 * The streams are NOT committed: this file regenerates them, byte for byte, wherever node runs (the build container and the GPU
 * box).  What is committed is what the REFERENCE made of them (tests/golden/corpus.json, written by
 * tests/golden/gen/gen_corpus.js in the build container): per stream the SHA-256 of the quantised spectra, the band words and
 * the unit records it parsed, a checksum and 64 probe samples of the PCM its readChunk() returned, and the message of the error
 * it threw, if any.
 *
 *   node tests/js/corpus_cases.js <outdir>     writes <name>.aac for every case and cases.json
 */
'use strict';
const fs = require('fs'), path = require('path');
const root = path.join(__dirname, '..', '..');
const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
const { Writer, BitWriter, Rng } = require('./aac_writer.js');
const { randomFrame, layoutChannels, PATTERN } = require('./stream_cases.js');

const LAYOUTS = [
    { tag: 'mono', layout: ['sce'] },
    { tag: 'stereo', layout: ['cpe'] },
    { tag: 'three', layout: ['sce', 'cpe'] },
    { tag: 'five1', layout: ['sce', 'cpe', 'cpe', 'lfe'] },
    { tag: 'seven', layout: ['sce', 'cpe', 'cpe', 'cpe'] },
    { tag: 'extras', layout: ['fil', 'dse', 'cpe', 'cce', 'dse', 'fil'] },
    { tag: 'coupled', layout: ['cce', 'sce', 'cce', 'cpe'] },
];
const VARIANTS = [
    { tag: 'plain', o: {} },
    { tag: 'tns', o: { tns: true } },
    { tag: 'split', o: { noCommon: true } },
];

/* the bad frames: name -> raw_data_block builder; the comment is the reference's throw site */
function malformed(wr, cb, rng) {
    const sce = function (mutate) { const ch = wr.randomChannel(rng, { seq: 0 }); mutate(ch); return wr.rawDataBlock([{ type: 'sce', id: 0, ch: [ch] }]); };
    const head = function (maxSfb, predictor) { const w = new BitWriter(); w.put(0, 3); w.put(0, 4); w.put(100, 8); w.put(0, 1); w.put(0, 2); w.put(0, 1); w.put(maxSfb, 6); w.put(predictor, 1); return w; };
    const good = function () { return wr.rawDataBlock([{ type: 'sce', id: 1, ch: [wr.randomChannel(rng, { seq: 0 })] }]); };
    return {
        band_type_12: function () { return sce(function (ch) { ch.bandTypes.fill(12); ch.sf.fill(ch.globalGain); }); },                                      // ics.js:96
        too_many_bands: function () { const w = head(10, 0); w.put(1, 4); w.put(11, 5); w.put(0, 32); return w.bytes(); },             // ics.js:105
        gain_control: function () { return sce(function (ch) { ch.gainControl = true; }); },                                      // ics.js:76
        pce: function () { return wr.rawDataBlock([{ type: 'pce', id: 0 }]); },                                                   // decoder.js:184
        prediction: function () { const w = head(10, 1); w.put(0, 32); return w.bytes(); },                                           // ics.js:318
        pulse_in_short: function () { const ch = wr.randomChannel(rng, { seq: 2 }); ch.pulse = { swb: 0, offset: [1], amp: [1] }; return wr.rawDataBlock([{ type: 'sce', id: 0, ch: [ch] }]); },   // ics.js:65
        pulse_data: function () { const ch = wr.randomChannel(rng, { seq: 0, pulse: true }); return wr.rawDataBlock([{ type: 'sce', id: 0, ch: [ch] }]); },   // ics.js:264
        truncated: function () { const b = good(); return b.subarray(0, b.length >> 1); },                                       // AV.Bitstream underflow
        ms_mask_3: function () { const w = new BitWriter(); w.put(1, 3); w.put(0, 4); w.put(1, 1); w.put(0, 1); w.put(0, 2); w.put(0, 1); w.put(10, 6); w.put(0, 1); w.put(3, 2); w.put(0, 32); return w.bytes(); },   // cpe.js:66
        scalefactor_range: function () { const w = head(1, 0); w.put(1, 4); w.put(1, 5); const e = cb.tables.sf[cb.tables.sf.length - 1]; w.put(e[1], e[0]); w.put(0, 64); return w.bytes(); },   // ics.js:165
    };
}

function cases() {
    const list = [];
    let k = 0;
    for (let si = 0; si < 12; si++)
        for (const L of LAYOUTS)
            for (const V of VARIANTS) {
                if (V.o.noCommon && L.layout.indexOf('cpe') < 0) continue;       // split windows need a pair
                list.push({ name: 'c' + String(k).padStart(3, '0') + '_' + L.tag + '_' + V.tag + '_si' + si, kind: 'clean', si: si, layout: L.layout, o: V.o, frames: 3 + (k % 3), seed: 0xC0FFEE + 7919 * k, phase: k });
                k++;
            }
    /* one bad frame behind two good ones, one stream per error and sample rate */
    const bad = ['band_type_12', 'too_many_bands', 'gain_control', 'pce', 'prediction', 'pulse_in_short', 'pulse_data', 'truncated', 'ms_mask_3', 'scalefactor_range'];
    for (let i = 0; i < bad.length; i++)
        for (const si of [3, 4, 8]) {
            list.push({ name: 'm' + String(k).padStart(3, '0') + '_' + bad[i] + '_si' + si, kind: 'malformed', bad: bad[i], si: si, layout: bad[i] === 'ms_mask_3' ? ['cpe'] : ['sce'], o: {}, frames: 3, seed: 0xBAD0000 + 104729 * k, phase: k });
            k++;
        }
    return list;
}

/* the ADTS frames of a case */
function build(c, cb) {
    cb = cb || codebooks.standard();
    const wr = new Writer(cb, c.si), rng = new Rng(c.seed >>> 0), C = layoutChannels(c.layout);
    const frames = [];
    const seqOf = function (t) { return function (ei) { return PATTERN[(t + ei + c.phase) % PATTERN.length]; }; };
    const wrap = function (body) {                                   /* Writer.adtsFrame around a ready-made raw_data_block */
        const w = new BitWriter();
        w.put(0xfff, 12); w.put(0, 1); w.put(0, 2); w.put(1, 1);
        w.put(1, 2); w.put(c.si, 4); w.put(0, 1); w.put(C & 7, 3); w.put(0, 4);
        w.put(7 + body.length, 13); w.put(0x7ff, 11); w.put(0, 2);
        const head = w.bytes(), out = new Uint8Array(head.length + body.length);
        out.set(head); out.set(body, head.length);
        return out;
    };
    const n = c.kind === 'malformed' ? c.frames - 1 : c.frames;
    for (let t = 0; t < n; t++) frames.push(wr.adtsFrame(randomFrame(wr, rng, c.layout, seqOf(t), c.o), C & 7, { crc: (c.phase + t) % 5 === 0 }));
    if (c.kind === 'malformed') frames.push(wrap(malformed(wr, cb, rng)[c.bad]()));
    return { frames: frames, C: C };
}

module.exports = { cases: cases, build: build, LAYOUTS: LAYOUTS };

if (require.main === module) {
    const outdir = process.argv[2];
    if (!outdir) { console.error('usage: corpus_cases.js <outdir>'); process.exit(2); }
    fs.mkdirSync(outdir, { recursive: true });
    const cb = codebooks.standard(), list = cases(), index = [];
    for (const c of list) {
        const s = build(c, cb);
        fs.writeFileSync(path.join(outdir, c.name + '.aac'), Buffer.concat(s.frames.map(function (b) { return Buffer.from(b); })));
        index.push({ name: c.name, kind: c.kind, si: c.si, channels: s.C, frames: s.frames.length, layout: c.layout });
    }
    fs.writeFileSync(path.join(outdir, 'cases.json'), JSON.stringify(index));
    console.log('corpus: ' + list.length + ' streams written to ' + outdir);
}
