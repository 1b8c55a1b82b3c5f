#!/usr/bin/env node
/*
 * tests/js/parse_cases.js — TEST KIT: inputs and expected outputs for the device front end (aacg_parse_*).
 *
 *   node tests/js/parse_cases.js <outdir> [standard|synthetic]
 *
 * Writes streams with the synthetic writer (aac_writer.js), parses them with the JavaScript front end (frontend.js,
 * itself checked against the reference's parser in test_frontend.js) and stores, per case, the bytes, the frame
 * table and what aacg_parse_batch must return for them: unit records, spectra, band words, TNS records, per-frame
 * results — in exactly the layout of include/aacgpu.h.  tests/test_parse_device.py compares the emulated kernel and
 * the GPU against these files bit for bit.
 *   standard (default)   the standard's codebooks as shipped (aac.js_amd/data/aac_codebooks.json)
 *   synthetic            stand-in codebooks (synth_codebooks.js: same alphabets, other prefix codes) — the table
 *                        builders and the parser must not depend on which prefix code they are given
 */
'use strict';
const fs = require('fs'), path = require('path');
const root = path.join(__dirname, '..', '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
const { FrontEnd } = require(path.join(root, 'aac.js_amd', 'js', 'frontend.js'));
const { Writer, BitWriter, Rng } = require('./aac_writer.js');
const { randomFrame, layoutChannels, CASES, PATTERN } = require('./stream_cases.js');
const { synthTables } = require('./synth_codebooks.js');

const outdir = process.argv[2], mode = process.argv[3] || 'standard';
const SALT = parseInt(process.env.AACG_CASE_SEED || '0', 10) >>> 0;      // other streams of the same kinds (tools/soak_parse.py)
if (!outdir) { console.error('usage: parse_cases.js <outdir> [standard|synthetic]'); process.exit(2); }
const cb = mode === 'synthetic' ? codebooks.fromTables(synthTables(0xC0DE)) : codebooks.standard();
fs.mkdirSync(outdir, { recursive: true });

/* the codebooks as aacg_code_entry records */
{
    const rec = cb.toEntryRecords();
    fs.writeFileSync(path.join(outdir, 'codebooks.entries'), Buffer.from(rec.entries));
    fs.writeFileSync(path.join(outdir, 'codebooks.counts'), Buffer.from(rec.counts.buffer));
}

const STATUS = [[/Insufficient data/, 1], [/Invalid band type/, 2], [/Too many bands/, 3], [/Scalefactor out of range/, 4], [/Pulse tool not allowed/, 5],
                [/Pulse (SWB|offset) out of range/, 6], [/TODO: add pulse data/, 7], [/TNS filter out of range/, 8], [/Prediction not implemented/, 9],
                [/gain control/, 10], [/PCE_ELEMENT/, 11], [/maxSFB out of range/, 12], [/Reserved ms mask/, 13], [/escape sequence too long/, 14]];
function statusOf(err) { for (const [re, code] of STATUS) if (re.test(err.message)) return code; throw err; }

const TYPE_CODE = { sce: 0, cpe: 1, lfe: 3 };
const manifest = [];

/* frames: [Uint8Array]; opts: { si, maxUnits, maxCh, applyPulses, quirks, wantTns } */
function emit(name, frames, o) {
    const fe = new FrontEnd({ codebooks: cb, referenceQuirks: o.quirks !== false }), config = { sampleIndex: o.si };
    const n = frames.length, blocks = n * o.maxCh;
    const table = new Uint32Array(2 * n), results = Buffer.alloc(8 * n);
    const unitBytes = new Uint8Array(n * o.maxUnits * host.UNIT_BYTES), q = new Int16Array(blocks * 1024), meta = new Uint16Array(blocks * 120);
    const tnsList = new Array(blocks).fill(null);
    const chunks = [];
    let offset = 0;
    frames.forEach(function (bytes, f) {
        if (f % 3 === 1) { chunks.push(Buffer.alloc(1 + (f % 5))); offset += 1 + (f % 5); }     // frames need not be adjacent or aligned
        table[2 * f] = offset; table[2 * f + 1] = bytes.length;
        chunks.push(Buffer.from(bytes)); offset += bytes.length;
        fe.pushPacket(bytes);
        let frame = null, status = 0;
        try {
            frame = fe.parseFrame({ config: config });
            for (const e of frame.elements) for (const c of e.ch) {
                if (c.pulse && !o.applyPulses) throw new Error('TODO: add pulse data');
                if (c.tns && o.wantTns) for (const w of c.tns.order) for (const ord of w) if (ord > 12) throw new Error('TNS filter out of range: ' + ord);
            }
        } catch (err) { status = statusOf(err); frame = null; }
        results.writeUInt8(status, 8 * f);
        if (!frame) return;
        let channel = 0, flags = 0;
        const units = [];
        if (frame.elements.length > o.maxUnits || frame.q.length / 1024 > o.maxCh) { results.writeUInt8(15, 8 * f); return; }
        frame.elements.forEach(function (e) {
            const block = f * o.maxCh + channel, anyTns = e.ch.some(function (c) { return !!c.tns; });
            e.ch.forEach(function (c, k) {
                const src = frame.q.subarray((channel + k) * 1024, (channel + k + 1) * 1024);
                if (c.pulse) host.applyPulses(src, c.pulse.offset, c.pulse.amp);
                q.set(src, (block + k) * 1024);
                meta.set(frame.meta.subarray((channel + k) * 120, (channel + k + 1) * 120), (block + k) * 120);
                if (c.tns && o.wantTns) { c.tns.short = c.windowSequence === 2; tnsList[block + k] = c.tns; }
                c.windowShapePrev = 0;
            });
            if (e.hasPns) flags |= 1;
            if (anyTns) flags |= 2;
            units.push({ stream: 0, pcmOffset: 0, channel: channel, nOutCh: 0, coefOffset: block, metaOffset: block, commonWindow: e.commonWindow,
                         maskPresent: e.maskPresent, hasPns: e.hasPns, ch: e.ch, tnsOffset: anyTns ? block : 0, tag: (TYPE_CODE[e.type] << 4) | e.id });
            channel += e.ch.length;
        });
        if (frame.hadCoupling) flags |= 4;                        // AACG_PARSE_HAS_CCE
        const packed = host.packUnits(units);
        units.forEach(function (u, i) { packed[i * host.UNIT_BYTES + 14] = u.tag; });            // reserved0: (element type << 4) | id
        unitBytes.set(packed, f * o.maxUnits * host.UNIT_BYTES);
        results.writeUInt8(units.length, 8 * f + 1); results.writeUInt8(channel, 8 * f + 2); results.writeUInt8(flags, 8 * f + 3);
        results.writeUInt32LE(frame.bitsUsed, 8 * f + 4);
    });
    const w = function (ext, data) { fs.writeFileSync(path.join(outdir, name + ext), data); };
    w('.bytes', Buffer.concat(chunks)); w('.frames', Buffer.from(table.buffer)); w('.units', Buffer.from(unitBytes));
    w('.q', Buffer.from(q.buffer)); w('.meta', Buffer.from(meta.buffer)); w('.results', results);
    if (o.wantTns) w('.tns', Buffer.from(host.packTns(tnsList)));
    manifest.push({ name: name, sampleIndex: o.si, frames: n, maxUnits: o.maxUnits, maxChannels: o.maxCh, wantTns: !!o.wantTns,
                    options: (o.applyPulses ? 1 : 0) | (o.quirks !== false ? 2 : 0) });
}

/* the stream cases of test_frontend.js, ADTS / ADTS+CRC / bare raw_data_block in turn */
for (const c of CASES) {
    const wr = new Writer(cb, c.si), rng = new Rng((0xF00D ^ (c.si * 7919) ^ c.frames ^ SALT) >>> 0), C = layoutChannels(c.layout), frames = [];
    for (let t = 0; t < c.frames; t++) {
        const elements = randomFrame(wr, rng, c.layout, function (ei) { return PATTERN[(t + ei) % PATTERN.length]; }, c.o);
        frames.push(t % 3 === 2 ? wr.rawDataBlock(elements) : wr.adtsFrame(elements, C & 7, { crc: t % 3 === 1 }));
    }
    const units = c.layout.filter(function (k) { return k === 'sce' || k === 'cpe' || k === 'lfe'; }).length;
    emit(c.name, frames, { si: c.si, maxUnits: units, maxCh: C, applyPulses: true, quirks: c.o.quirks, wantTns: !!c.o.tns });
    if (c.o.pulse) emit(c.name + '_nopulse', frames, { si: c.si, maxUnits: units + 1, maxCh: C + 1, applyPulses: false, quirks: c.o.quirks, wantTns: false });
}

/* a larger batch: more frames than one workgroup has lanes */
{
    const wr = new Writer(cb, 3), rng = new Rng((0xB16 ^ SALT) >>> 0), frames = [];
    for (let t = 0; t < 600; t++) frames.push(wr.adtsFrame(randomFrame(wr, rng, ['cpe'], function () { return PATTERN[t % PATTERN.length]; }, { tns: t % 4 === 0 }), 2));
    emit('stereo600', frames, { si: 3, maxUnits: 1, maxCh: 2, wantTns: true });
}

/* malformed frames between good ones: every status code */
{
    const wr = new Writer(cb, 3), rng = new Rng(99), frames = [];
    const good = function () { return wr.rawDataBlock([{ type: 'sce', id: 1, ch: [wr.randomChannel(rng, { seq: rng.below(4) })] }]); };
    const sce = function (mutate) { const ch = wr.randomChannel(rng, { seq: 0 }); mutate(ch); return wr.rawDataBlock([{ type: 'sce', id: 0, ch: [ch] }]); };
    const head = function (maxSfb, predictor) { const w = new BitWriter(); w.put(0, 3); w.put(0, 4); w.put(100, 8); w.put(0, 1); w.put(0, 2); w.put(0, 1); w.put(maxSfb, 6); w.put(predictor, 1); return w; };
    frames.push(good());
    frames.push(sce(function (ch) { ch.bandTypes.fill(12); }));                                      // 2
    { const w = head(10, 0); w.put(1, 4); w.put(11, 5); w.put(0, 32); frames.push(w.bytes()); }     // 3
    frames.push(sce(function (ch) { ch.gainControl = true; }));                                       // 10
    frames.push(wr.rawDataBlock([{ type: 'pce', id: 0 }]));                                           // 11
    { const w = head(50, 0); w.put(0, 32); frames.push(w.bytes()); }                                 // 12
    { const w = head(10, 1); w.put(0, 32); frames.push(w.bytes()); }                                 // 9
    { const ch = wr.randomChannel(rng, { seq: 2 }); ch.pulse = { swb: 0, offset: [1], amp: [1] }; frames.push(wr.rawDataBlock([{ type: 'sce', id: 0, ch: [ch] }])); }   // 5
    frames.push(good());
    { const b = good(); frames.push(b.subarray(0, b.length >> 1)); }                                  // 1
    { const w = new BitWriter(); w.put(1, 3); w.put(0, 4); w.put(1, 1); w.put(0, 1); w.put(0, 2); w.put(0, 1); w.put(10, 6); w.put(0, 1); w.put(3, 2); w.put(0, 32); frames.push(w.bytes()); }   // 13
    { const w = head(1, 0); w.put(1, 4); w.put(1, 5); w.put(cb.tables.sf[cb.tables.sf.length - 1][1], cb.tables.sf[cb.tables.sf.length - 1][0]);   // one band, extreme scalefactor step
      w.put(0, 64); frames.push(w.bytes()); }
    frames.push(wr.rawDataBlock([{ type: 'sce', id: 0, ch: [wr.randomChannel(rng, { seq: 0 })] }, { type: 'sce', id: 1, ch: [wr.randomChannel(rng, { seq: 0 })] }]));   // 15: two elements, one allowed
    frames.push(good());
    emit('malformed', frames, { si: 3, maxUnits: 1, maxCh: 1, applyPulses: true });
}
/* garbage in: random bytes, and valid frames with bits flipped — whatever the JavaScript parser makes of them (which
 * status, and for the frames that still parse, which records), the device parser must make the same of them */
{
    const wr = new Writer(cb, 4), rng = new Rng((0xBADF00D ^ SALT) >>> 0), frames = [];
    for (let t = 0; t < 200; t++) {
        const b = new Uint8Array(20 + rng.below(700));
        for (let i = 0; i < b.length; i++) b[i] = rng.below(256);
        if (t % 4 === 0) b[0] &= 0x1f;                            // starts like an SCE: gets further into the syntax
        if (t % 4 === 1) { b[0] = 0x20 | (b[0] & 0x1f); }         // starts like a CPE
        frames.push(b);
    }
    for (let t = 0; t < 300; t++) {
        const layout = t % 3 === 0 ? ['sce'] : t % 3 === 1 ? ['cpe'] : ['sce', 'cpe'];
        const b = wr.rawDataBlock(randomFrame(wr, rng, layout, function (ei) { return PATTERN[(t + ei) % PATTERN.length]; }, { tns: true, pns: true, pulse: t % 5 === 0 })).slice();
        for (let flips = 1 + rng.below(3); flips > 0; flips--) b[rng.below(b.length)] ^= 1 << rng.below(8);
        frames.push(b);
    }
    emit('fuzz', frames, { si: 4, maxUnits: 2, maxCh: 3, applyPulses: true, wantTns: true });
}
fs.writeFileSync(path.join(outdir, 'manifest.json'), JSON.stringify(manifest));
console.log('parse cases written: ' + manifest.length + ' (' + mode + ' codebooks)');
