#!/usr/bin/env node
/*
 * tests/golden/gen/gen_corpus.js — BUILD CONTAINER ONLY: runs the REAL reference (/root/reference/src) on every stream of
 * tests/js/corpus_cases.js and writes what it made of them to tests/golden/corpus.json (small: hashes, a checksum, probes).
 *
 *   node tests/golden/gen/gen_corpus.js            regenerate tests/golden/corpus.json
 *   node tests/golden/gen/gen_corpus.js --check    regenerate in memory and compare with the committed file
 *
 * Per stream, frame by frame through the reference's own AACDecoder.readChunk() (src/decoder.js:125-216: its Bitstream calls,
 * Huffman, ICS / CPE / CCE parsers, process(), interleave) until it throws:
 *   - what it PARSED: per frame the elements it pushed (decoder.js:138-166) — window info, band types, scalefactors — and every
 *     integer its Huffman decoder returned (ics.js:247).  The repo's JavaScript front end parses the same bytes; the two must
 *     agree field by field and integer by integer (asserted here), and then the front end's engine records — quantised spectra,
 *     band words, unit records: the layout the device parser writes — are hashed (SHA-256).
 *   - what it DECODED: the Float32Array every readChunk() returned, as sum, sum of squares and 64 probe samples.
 *   - the message of the error it threw, and at which frame.
 * No reference source is copied: the reference is require()'d where it lies and only its outputs are kept.
 */
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto'), assert = require('assert');
const root = path.join(__dirname, '..', '..', '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const { BitStream } = require(path.join(root, 'aac.js_amd', 'js', 'bits.js'));
const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
const { FrontEnd, swbOffsets } = require(path.join(root, 'aac.js_amd', 'js', 'frontend.js'));
const corpus = require(path.join(root, 'tests', 'js', 'corpus_cases.js'));

const REF = '/root/reference/src/';
if (!fs.existsSync(REF + 'decoder.js')) { console.error('gen_corpus.js needs the reference checkout at ' + REF); process.exit(2); }
process.env.NODE_PATH = path.join(root, 'tests', 'golden', 'gen', 'stubs');
require('module').Module._initPaths();
const AACDecoder = require(REF + 'decoder.js'), FilterBank = require(REF + 'filter_bank.js'), refTables = require(REF + 'tables.js'), refHuffman = require(REF + 'huffman.js');
let ints = null;
const inner = refHuffman.decodeSpectralData;
refHuffman.decodeSpectralData = function (stream, book, data, off) {               // ics.js:247 calls through the module object
    inner.call(this, stream, book, data, off);
    for (let j = 0; j < (book < 5 ? 4 : 2); j++) ints.push(data[off + j]);
};

/* SHA-256, base64 (43 characters instead of 64: the file stays under 200 KB) */
const sha = function (buf) { return crypto.createHash('sha256').update(buf).digest('base64').replace(/=+$/, ''); };
/* the probe positions: the same integer arithmetic in tests/test_corpus.py */
function probeIndex(k, n) { return ((k * 7919 + 13) * 104729) % n; }
/* a unit record as the device parser and the front end both produce it, without the fields a caller fills in (stream, PCM and
 * block offsets, TNS flags): n_ch, flags, channel, and per channel window sequence / shape / max_sfb / grouping */
function canonicalUnits(packed) {
    const n = packed.length / host.UNIT_BYTES, out = [];
    for (let i = 0; i < n; i++) {
        const o = i * host.UNIT_BYTES, nch = packed[o + 12];
        out.push(nch, packed[o + 13], packed[o + 8], packed[o + 9]);
        for (let c = 0; c < nch; c++) {
            const b = o + 24 + 16 * c;
            out.push(packed[b], packed[b + 1], packed[b + 3], packed[b + 4]);
            for (let g = 0; g < 8; g++) out.push(packed[b + 8 + g]);
        }
    }
    return Buffer.from(out);
}

const cb = codebooks.standard();
const entries = [];
const messages = {};
for (const c of corpus.cases()) {
    const s = corpus.build(c, cb), C = s.C;
    const ref = new AACDecoder();
    ref.format = {};
    ref.config = { profile: 2, sampleIndex: c.si, sampleRate: host.SAMPLE_RATES[c.si], chanConfig: C, frameLength: 1024 };   // what setCookie leaves (decoder.js:53-113)
    ref.filter_bank = new FilterBank(false, C);                                                                                // decoder.js:112
    let refElements = null;
    const process0 = AACDecoder.prototype.process;
    ref.process = function (elements) { refElements = elements; process0.call(this, elements); };
    /* the repo's front end on the same bytes, with an engine that records what it would be given */
    const captured = {};
    const engine = { resetStream: function () {}, decodeBatch: function (units, q, meta) { captured.units = units; captured.q = q; captured.meta = meta; } };
    const dec = new host.GpuAACDecoder({ engine: engine, frontend: new FrontEnd({ codebooks: cb }), lookahead: s.frames.length, applyPulses: false });
    dec.config = { profile: 2, sampleIndex: c.si, chanConfig: C };
    const pcm = [];
    let error = null, good = 0;
    for (let t = 0; t < s.frames.length && !error; t++) {
        ints = [];
        ref.bitstream = new BitStream(s.frames[t]);
        refElements = null;
        let out = null;
        try { out = ref.readChunk(); } catch (e) { error = { frame: t, message: String(e.message) }; }
        if (!out) break;
        assert.strictEqual(ref.bitstream.pos, s.frames[t].length * 8, c.name + ': the reference did not end on the frame boundary');
        pcm.push(out);
        good++;
        /* the front end's parse of this frame against the reference's */
        const fe1 = new FrontEnd({ codebooks: cb });
        fe1.push(s.frames[t]);
        const f = fe1.parseFrame(dec);
        assert.strictEqual(refElements.length, f.elements.length, c.name + ' frame ' + t + ': elements');
        let block = 0;
        const want = [];
        refElements.forEach(function (re, i) {
            const me = f.elements[i], chans = re.left ? [re.left, re.right] : [re];
            if (re.left) { assert.strictEqual(!!re.commonWindow, me.commonWindow); assert.strictEqual(!!re.maskPresent, me.maskPresent); }
            chans.forEach(function (ics, ci) {
                const mc = me.ch[ci], info = ics.info, nb = info.groupCount * info.maxSFB, words = f.meta.subarray(block * 120, block * 120 + 120);
                assert.deepStrictEqual([info.windowSequence, info.windowShape[1], info.maxSFB, Array.from(info.groupLength.subarray(0, info.groupCount))],
                                       [mc.windowSequence, mc.windowShape, mc.maxSFB, mc.groupLength], c.name + ': ics_info');
                for (let b = 0; b < nb; b++) {
                    assert.strictEqual(ics.bandTypes[b], words[b] >>> 12, c.name + ': band type');
                    const sf = ics.bandTypes[b] ? refTables.SCALEFACTOR_TABLE[words[b] & 0x1ff] * (words[b] & 0x200 ? -1 : 1) : 0;
                    assert.strictEqual(ics.scaleFactors[b], sf, c.name + ': scalefactor of band ' + b);
                    if (re.left && ci === 0 && re.maskPresent) assert.strictEqual(!!re.ms_used[b], !!(words[b] & 0x400), c.name + ': ms_used');
                }
                const off = swbOffsets(c.si, mc.windowSequence === 2), q = f.q.subarray(block * 1024, block * 1024 + 1024);
                for (let g = 0, groupOff = 0; g < mc.groupLength.length; groupOff += mc.groupLength[g] * 128, g++)
                    for (let sfb = 0; sfb < mc.maxSFB; sfb++) {
                        const bt = words[g * mc.maxSFB + sfb] >>> 12;
                        if (bt === 0 || bt >= 13) continue;
                        for (let w = 0; w < mc.groupLength[g]; w++) for (let k = off[sfb]; k < off[sfb + 1]; k++) want.push(q[groupOff + w * 128 + k]);
                    }
                block++;
            });
        });
        /* every integer the reference's Huffman decoder returned (coupling elements' integers are in its log too, not in mine) */
        if (c.layout.indexOf('cce') < 0) assert.deepStrictEqual(ints, want, c.name + ' frame ' + t + ': Huffman output');
    }
    const e = { name: c.name, si: c.si, channels: C, frames: s.frames.length, decoded: good, error: error, bytes: sha(Buffer.concat(s.frames.map(function (b) { return Buffer.from(b); }))).slice(0, 16) };
    if (error) messages[error.message] = (messages[error.message] || 0) + 1;
    if (good) {
        /* the engine records of the frames that decoded: the whole prefix through the plugin surface, one batch */
        const fe = dec.frontend;
        for (let t = 0; t < good; t++) fe.push(s.frames[t]);
        dec.lookahead = good;
        dec.readChunk();
        e.units = sha(canonicalUnits(captured.units));
        e.q = sha(Buffer.from(captured.q.buffer, captured.q.byteOffset, captured.q.byteLength));
        e.meta = sha(Buffer.from(captured.meta.buffer, captured.meta.byteOffset, captured.meta.byteLength));
        e.n_units = captured.units.length / host.UNIT_BYTES;
        const all = new Float32Array(good * 1024 * C);
        pcm.forEach(function (p, t) { all.set(p, t * 1024 * C); });
        let sum = 0, sumsq = 0;
        for (let i = 0; i < all.length; i++) { sum += all[i]; sumsq += all[i] * all[i]; }
        const probes = new Float32Array(64);
        for (let k = 0; k < 64; k++) probes[k] = all[probeIndex(k, all.length)];
        e.pcm = { n: all.length, sum: sum, sumsq: sumsq, probes: Buffer.from(probes.buffer).toString('base64') };
    }
    entries.push(e);
}
const text = JSON.stringify({ generator: 'tests/golden/gen/gen_corpus.js against /root/reference/src (aac.js 0.1.3)', streams: entries }) + '\n';
const file = path.join(root, 'tests', 'golden', 'corpus.json');
if (process.argv[2] === '--check') {
    assert.strictEqual(fs.readFileSync(file, 'utf8'), text, 'tests/golden/corpus.json is not what the reference produces now');
    console.log('corpus check ok: ' + entries.length + ' streams');
} else {
    fs.writeFileSync(file, text);
    console.log('corpus: ' + entries.length + ' streams, ' + entries.reduce(function (a, e) { return a + e.decoded; }, 0) + ' frames decoded by the reference, ' + text.length + ' bytes');
    console.log('errors the reference threw: ' + JSON.stringify(messages, null, 1));
}
