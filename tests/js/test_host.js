#!/usr/bin/env node
/* Node-side tests of the JavaScript host.  `node tests/js/test_host.js cpu` needs no GPU;
 * `... gpu` drives the engine through the N-API addon against the golden vectors. */
'use strict';
const fs = require('fs'), path = require('path'), assert = require('assert');
const root = path.join(__dirname, '..', '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));

function loadGolden() {
    const man = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'golden.json')));
    const blob = fs.readFileSync(path.join(root, 'tests', 'golden', 'golden.bin'));
    const T = { f4: Float32Array, f8: Float64Array, i4: Int32Array, i2: Int16Array, u2: Uint16Array, u1: Uint8Array };
    const out = {};
    for (const name of Object.keys(man.arrays)) {
        const a = man.arrays[name], n = a.shape.reduce((x, y) => x * y, 1), Ty = T[a.dtype];
        const copy = new Uint8Array(n * Ty.BYTES_PER_ELEMENT);
        copy.set(blob.subarray(a.offset, a.offset + copy.length));
        out[name] = new Ty(copy.buffer);
    }
    return out;
}
function rms(a, b) { let s = 0; for (let i = 0; i < a.length; i++) s += (a[i] - b[i]) * (a[i] - b[i]); return Math.sqrt(s / a.length); }

const mode = process.argv[2] || 'cpu';
const g = loadGolden();

// unit packing round-trips the records the reference-derived fixtures hold
for (const name of ['scn_stereo', 'scn_7ch', 'scn_split', 'scn_mono']) {
    const bytes = g[name + '.units'];
    const again = host.packUnits(host.unpackUnits(bytes));
    assert.deepStrictEqual(Buffer.from(again), Buffer.from(bytes), name + ': packUnits(unpackUnits(x)) != x');
}
assert.strictEqual(host.packBandWord(15, 200, false, true), (15 << 12) | 0x400 | 200);

// setCookie: 2-byte AudioSpecificConfig as the ADTS demuxer synthesises it (adts_demuxer.js:67-70): LC, 48 kHz, stereo
{
    const dec = new host.GpuAACDecoder({ engine: { resetStream: function () {} } });
    dec.init();
    assert.strictEqual(dec.format.floatingPoint, true);
    dec.setCookie(new Uint8Array([(2 << 3) | (3 >> 1), ((3 & 1) << 7) | (2 << 3)]));
    assert.deepStrictEqual([dec.config.profile, dec.config.sampleIndex, dec.config.sampleRate, dec.config.chanConfig, dec.config.frameLength],
                           [2, 3, 48000, 2, 1024]);
    assert.strictEqual(dec.format.channelsPerFrame, 2);
    assert.throws(() => dec.setCookie(new Uint8Array([(5 << 3) | 1, 0x90])), /not supported/);
}
// TNS side info: aacg_tns_info layout (include/aacgpu.h): n_filt[8], then 8 x {length, order, direction, pad, coef[12]}
const tnsLong = { short: false, nFilt: [2], length: [[20, 9]], order: [[3, 1]], direction: [[false, true]],
                  coef: [[new Float32Array([0.5, -0.25, 0.125]), new Float32Array([-0.4338837])]] };
{
    const tnsShort = { short: true, nFilt: [0, 1, 0, 0, 0, 0, 0, 1], length: [[], [7], [], [], [], [], [], [5]], order: [[], [2], [], [], [], [], [], [0]],
                       direction: [[], [true], [], [], [], [], [], [false]], coef: [[], [new Float32Array([0.25, 0.75])], [], [], [], [], [], [new Float32Array(0)]] };
    const bytes = host.packTns([tnsLong, null, tnsShort]);
    assert.strictEqual(bytes.length, 3 * host.TNS_BYTES);
    const v = new DataView(bytes.buffer);
    assert.deepStrictEqual([v.getUint8(0), v.getUint8(8), v.getUint8(9), v.getUint8(10), v.getUint8(60), v.getUint8(61), v.getUint8(62)],
                           [2, 20, 3, 0, 9, 1, 1]);
    assert.strictEqual(v.getFloat32(12 + 4, true), -0.25);
    assert.strictEqual(v.getFloat32(64, true), Math.fround(-0.4338837));
    for (let i = host.TNS_BYTES; i < 2 * host.TNS_BYTES; i++) assert.strictEqual(bytes[i], 0);       // channel without TNS
    const o = 2 * host.TNS_BYTES;
    assert.deepStrictEqual([v.getUint8(o + 1), v.getUint8(o + 7), v.getUint8(o + 8 + 52), v.getUint8(o + 8 + 52 + 1), v.getUint8(o + 8 + 52 + 2)], [1, 1, 7, 2, 1]);
    assert.strictEqual(v.getFloat32(o + 8 + 52 + 8, true), 0.75);
    assert.throws(() => host.packTns([{ short: false, nFilt: [1], length: [[1]], order: [[13]], direction: [[0]], coef: [[new Float32Array(13)]] }]), /out of range/);
    // unit records carry the flag and the record index
    const u = host.unpackUnits(g['scn_stereo.units'])[0];
    u.ch[1].tns = tnsLong; u.tnsOffset = 6;
    const ub = host.packUnits([u]);
    assert.deepStrictEqual([ub[24 + 5], ub[40 + 5], new DataView(ub.buffer).getUint32(56, true)], [0, 1, 6]);
}
// pulse data: away from zero on the quantised integers; the reference's behaviour (throw) is the default
{
    const q = new Int16Array(1024); q[10] = 3; q[11] = -2; q[40] = 0;
    host.applyPulses(q, [10, 11, 40], [5, 7, 1]);
    assert.deepStrictEqual([q[10], q[11], q[40]], [8, -9, -1]);                // q == 0 moves down, like the standard's `else` branch
    assert.throws(() => host.applyPulses(q, [1024], [1]), /Pulse offset out of range/);
    const mk = function (applyPulses) {
        const fq = new Int16Array(2048); fq[1024 + 7] = 4;
        const dec = new host.GpuAACDecoder({ engine: { resetStream: function () {} }, applyPulses: applyPulses });
        dec.setCookie(new Uint8Array([(2 << 3) | (3 >> 1), ((3 & 1) << 7) | (2 << 3)]));
        const ch = function (pulse) { return { windowSequence: 0, windowShape: 1, maxSFB: 49, groupLength: [1], pulse: pulse }; };
        const frame = { elements: [{ type: 'cpe', commonWindow: true, maskPresent: false, ch: [ch(null), ch({ offset: [7], amp: [3] })] }], q: fq, meta: new Uint16Array(240) };
        dec.unitsOfFrame(frame, 0, 0, null);
        return fq[1024 + 7];
    };
    assert.strictEqual(mk(true), 7);
    assert.throws(() => mk(false), /TODO: add pulse data/);
}
// readChunk look-ahead: a malformed frame in the middle of a batch is thrown by the call that reaches it — after the good
// frames before it have been returned, and the frames behind it still follow (decoder.js:125-201 throws at exactly that
// frame); a frame the host itself refuses (noise bands without PNS_SPEC, pulse data) behaves the same way
{
    const frame = function (extra) {
        const ch = Object.assign({ windowSequence: 0, windowShape: 0, maxSFB: 0, groupLength: [1] }, (extra && extra.ch) || {});
        return { elements: [Object.assign({ type: 'sce', id: 0, ch: [ch] }, (extra && extra.el) || {})], q: new Int16Array(1024), meta: new Uint16Array(120) };
    };
    const script = [frame(), frame(), frame(), new Error('Invalid band type: 12'), frame(), frame({ el: { hasPns: true } }), frame(),
                    frame({ ch: { pulse: { offset: [1], amp: [1] } } }), frame(), null, null];
    let at = 0, batches = [];
    const frontend = { parseFrame: function () { const s = script[at++]; if (s instanceof Error) throw s; return s; } };
    const engine = { resetStream: function () {}, decodeBatch: function (units, q, meta, pcm) { batches.push(units.length / host.UNIT_BYTES); pcm.fill(batches.length); } };
    const dec = new host.GpuAACDecoder({ engine: engine, frontend: frontend, lookahead: 16 });
    dec.config = { profile: 2, sampleIndex: 3, chanConfig: 1 };
    const seen = [];
    for (let i = 0; i < 11; i++) { try { const r = dec.readChunk(); seen.push(r ? r[0] : null); } catch (e) { seen.push(e.message.replace(/^aacgpu: (NOISE_BT).*/, '$1')); } }
    assert.deepStrictEqual(seen, [1, 1, 1, 'Invalid band type: 12', 2, 'NOISE_BT', 3, 'TODO: add pulse data', 4, null, null]);
    assert.deepStrictEqual(batches, [3, 1, 1, 1]);
}
// look-ahead batches end where the set of elements changes (an element that appears or disappears mid-stream — a coupling
// element with its own filterbank included — is a new chain for the planner), and an engine error takes the place of its
// batch's frames in the queue: later frames still follow, in order
{
    const frame = function (types) {
        return { elements: types.map(function (t, i) { return { type: t, id: (i * 7 + types.length) & 15, ch: [{ windowSequence: 0, windowShape: 0, maxSFB: 0, groupLength: [1] }] }; }),
                 q: new Int16Array(1024 * types.length), meta: new Uint16Array(120 * types.length) };
    };
    const script = [frame(['sce']), frame(['sce']), frame(['sce', 'lfe']), frame(['sce', 'lfe']), frame(['sce']), frame(['sce']), frame(['sce']), null, null];
    let at = 0, batches = [];
    const frontend = { parseFrame: function () { return script[at++]; } };
    const engine = { resetStream: function () {}, decodeBatch: function (units, q, meta, pcm) {
        batches.push(units.length / host.UNIT_BYTES);
        if (batches.length === 2) throw new Error('aacgpu: aacg_decode_batch failed (-7)');
        pcm.fill(batches.length);
    } };
    const dec = new host.GpuAACDecoder({ engine: engine, frontend: frontend, lookahead: 16 });
    dec.config = { profile: 2, sampleIndex: 3, chanConfig: 2 };
    const seen = [];
    for (let i = 0; i < 7; i++) { try { const r = dec.readChunk(); seen.push(r ? r[0] : null); } catch (e) { seen.push('error'); } }
    assert.deepStrictEqual(batches, [2, 4, 3]);               // units per batch: 2 frames x 1, 2 frames x 2, 3 frames x 1
    assert.deepStrictEqual(seen, [1, 1, 'error', 3, 3, 3, null]);
}
// ADTS framing (aac.js_amd/js/adts.js): headers built here bit by bit from the field layout
{
    const adts = require(path.join(__dirname, '..', '..', 'aac.js_amd', 'js', 'adts.js'));
    const header = function (profile, sr, ch, frameLength, crc) {
        const bits = [];
        const put = function (v, n) { for (let i = n - 1; i >= 0; i--) bits.push((v >> i) & 1); };
        put(0xfff, 12); put(0, 1); put(0, 2); put(crc ? 0 : 1, 1); put(profile - 1, 2); put(sr, 4); put(0, 1); put(ch, 3);
        put(0, 4); put(frameLength, 13); put(0x7ff, 11); put(0, 2);
        if (crc) put(0xbeef, 16);
        const b = new Uint8Array(bits.length / 8);
        bits.forEach(function (x, i) { b[i >> 3] |= x << (7 - (i & 7)); });
        return b;
    };
    const frame = function (profile, sr, ch, payload, crc) {
        const h = header(profile, sr, ch, (crc ? 9 : 7) + payload, crc), f = new Uint8Array(h.length + payload);
        f.set(h); for (let i = h.length; i < f.length; i++) f[i] = (i * 37) & 0x7f;          // never looks like a syncword
        return f;
    };
    const h = adts.readHeader(new host.BitReader(header(2, 3, 2, 345, false)));
    assert.deepStrictEqual(h, { profile: 2, samplingIndex: 3, chanConfig: 2, frameLength: 345, numFrames: 1, headerBytes: 7 });
    const r = new host.BitReader(header(2, 4, 6, 100, true));
    assert.strictEqual(adts.readHeader(r).headerBytes, 9);
    assert.strictEqual(r.pos, 72);                                                           // the CRC is consumed too
    assert.throws(() => adts.readHeader(new host.BitReader(new Uint8Array([0xff, 0xe1, 0, 0, 0, 0, 0]))), /Invalid ADTS header/);
    // the synthesised cookie parses back to the same configuration (adts_demuxer.js:66-70 -> decoder.js:53-113)
    const dec = new host.GpuAACDecoder({ engine: { resetStream: function () {} } });
    dec.setCookie(adts.cookie(h));
    assert.deepStrictEqual([dec.config.profile, dec.config.sampleIndex, dec.config.chanConfig], [2, 3, 2]);
    // probe: syncword at an even offset only, like the reference's readUInt16 loop
    assert.strictEqual(adts.probe(frame(2, 3, 2, 20, false)), true);
    assert.strictEqual(adts.probe(new Uint8Array([0, 0xff, 0xf1, 0, 0, 0])), false);
    assert.strictEqual(adts.probe(new Uint8Array([0, 0, 0xff, 0xf1, 0, 0])), true);
    assert.strictEqual(adts.probe(new Uint8Array([1, 2, 3, 4])), false);
    // frames(): complete frames only
    const f1 = frame(2, 3, 2, 50, false), f2 = frame(2, 3, 2, 70, true), f3 = frame(2, 3, 2, 30, false);
    const buf = new Uint8Array(f1.length + f2.length + f3.length - 5);
    buf.set(f1); buf.set(f2, f1.length); buf.set(f3.subarray(0, f3.length - 5), f1.length + f2.length);
    const fr = adts.frames(buf);
    assert.deepStrictEqual(fr.map(function (x) { return [x.offset, x.length]; }), [[0, 57], [57, 79]]);
    // demuxer events: format, cookie, data — once, then data only
    const ev = [];
    const dm = new adts.AdtsDemuxer(function (name, payload) { ev.push([name, payload]); });
    dm.push(f1); dm.push(f2);
    assert.deepStrictEqual(ev.map(function (e) { return e[0]; }), ['format', 'cookie', 'data', 'data']);
    assert.deepStrictEqual(ev[0][1], { formatID: 'aac ', sampleRate: 48000, channelsPerFrame: 2, bitsPerChannel: 16 });
    assert.deepStrictEqual(Array.from(ev[1][1]), [17, 144]);
}
console.log('host cpu tests ok');
if (mode !== 'gpu') process.exit(0);

// ---- GPU: raw batch through the addon, then the plugin surface with look-ahead ------------------
for (const name of ['scn_stereo', 'scn_7ch']) {
    const C = name === 'scn_7ch' ? 7 : 2, ref = g[name + '.pcm'];
    const eng = new host.Engine({ maxStreams: 1, maxChannels: C });
    const pcm = new Float32Array(ref.length);
    eng.decodeBatch(g[name + '.units'], g[name + '.q'], g[name + '.meta'], pcm);
    const e = rms(pcm, ref);
    assert.ok(e < 1e-5, name + ' rms ' + e);
    console.log(name, 'decodeBatch rms', e.toExponential(3));
}
{
    // readChunk(): frames arrive one at a time from a front end; look-ahead 5 batches them
    const name = 'scn_stereo', units = host.unpackUnits(g[name + '.units']), ref = g[name + '.pcm'];
    let t = 0, avail = 0;
    const frontend = { parseFrame: function () {
        if (t >= units.length || t >= avail) return null;
        const u = units[t], q = g[name + '.q'].subarray(2048 * t, 2048 * (t + 1)), meta = g[name + '.meta'].subarray(240 * t, 240 * (t + 1));
        t++;
        return { elements: [{ type: 'cpe', commonWindow: u.commonWindow, maskPresent: u.maskPresent, ch: u.ch }], q: q, meta: meta };
    } };
    const dec = new host.GpuAACDecoder({ frontend: frontend, lookahead: 5 });
    dec.init();
    dec.setCookie(new Uint8Array([(2 << 3) | (3 >> 1), ((3 & 1) << 7) | (2 << 3)]));
    let worst = 0, got = 0;
    for (const burst of [1, 7, 3, 7]) {          // data arrives in bursts: 18 frames in total
        avail += burst;
        for (let out; (out = dec.readChunk()) !== null; got++) {
            assert.strictEqual(out.length, 2048);
            worst = Math.max(worst, rms(out, ref.subarray(2048 * got, 2048 * (got + 1))));
        }
    }
    assert.strictEqual(got, 18);
    assert.ok(worst < 1e-5, 'readChunk rms ' + worst);
    console.log('readChunk x18 worst frame rms', worst.toExponential(3));
    // error convention: a bad batch throws, like the reference's throw new Error(...)
    assert.throws(() => dec.engine.decodeBatch(g[name + '.units'], new Int16Array(1024), g[name + '.meta'], new Float32Array(2048)), /aacgpu/);
}
{
    // TNS_SPEC engine: side info reaches the kernels; a TNS_REFERENCE engine ignores it (numerical parity: tests/test_tns_spec.py)
    const name = 'scn_stereo', ref = g[name + '.pcm'], units = host.unpackUnits(g[name + '.units']);
    const list = [];
    units.forEach(function (u, i) {
        if (u.ch[0].windowSequence === 2) return;
        u.ch[0].tns = tnsLong; u.tnsOffset = list.length; list.push(tnsLong, null);
    });
    const ub = host.packUnits(units), tb = host.packTns(list);
    const plain = new host.Engine({ maxStreams: 1, maxChannels: 2 }), a = new Float32Array(ref.length);
    plain.decodeBatch(ub, g[name + '.q'], g[name + '.meta'], a, tb);
    assert.ok(rms(a, ref) < 1e-5, 'TNS_REFERENCE engine must ignore TNS side info');
    const spec = new host.Engine({ maxStreams: 1, maxChannels: 2, tnsMode: host.TNS_SPEC }), b = new Float32Array(ref.length);
    spec.decodeBatch(ub, g[name + '.q'], g[name + '.meta'], b, tb);
    assert.ok(b.every(Number.isFinite) && rms(b, ref) > 1e-4, 'TNS_SPEC engine must apply the filters');
    console.log('tns side info ok, rms vs identity', rms(b, ref).toExponential(3));
}
{
    // GpuFrontEnd: the same frames parsed on the device and by the JavaScript front end (the standard's codebooks), then
    // decoded through GpuAACDecoder.readChunk with either front end: identical frame objects, identical PCM
    const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
    const { Writer, Rng } = require('./aac_writer.js'), { randomFrame, PATTERN } = require('./stream_cases.js'), { synthTables } = require('./synth_codebooks.js');
    const cb = process.env.AACG_TEST_STANDIN_BOOKS ? codebooks.fromTables(synthTables(0xC0DE)) : codebooks.standard(), wr = new Writer(cb, 3), rng = new Rng(0x6F0);
    const frames = [];
    for (let t = 0; t < 40; t++) frames.push(wr.adtsFrame(randomFrame(wr, rng, ['cpe'], function () { return PATTERN[t % PATTERN.length]; }, { tns: t % 3 === 0 }), 2, { crc: t % 2 === 1 }));
    const bad = frames[7].slice(); bad[9] ^= 0x5a; bad[15] ^= 0xff;      // a damaged frame: both front ends must agree on what happens
    const stream = Buffer.concat(frames.map(function (b) { return Buffer.from(b); }));
    const cpu = new host.FrontEnd({ codebooks: cb }), gpu = new host.GpuFrontEnd({ codebooks: cb, wantTns: true, batch: 16 });
    const config = { sampleIndex: 3 };
    cpu.push(new Uint8Array(stream)); gpu.push(new Uint8Array(stream));
    cpu.pushPacket(bad); gpu.pushPacket(bad);
    let n = 0;
    for (;;) {
        let a = null, b = null, ea = null, eb = null;
        try { a = cpu.parseFrame({ config: config }); } catch (e) { ea = e.message.replace(/\s*[:(].*$/, ''); }
        try { b = gpu.parseFrame({ config: config }); } catch (e) { eb = e.message.replace(/\s*[:(].*$/, ''); }
        assert.strictEqual(eb, ea, 'frame ' + n + ': error behaviour');
        if (!a && !ea) { assert.strictEqual(b, null); break; }
        if (a) {
            assert.deepStrictEqual(b.q, a.q, 'frame ' + n + ': spectrum'); assert.deepStrictEqual(b.meta, a.meta, 'frame ' + n + ': band words');
            assert.strictEqual(b.elements.length, a.elements.length);
            a.elements.forEach(function (e, i) {
                const x = b.elements[i];
                assert.deepStrictEqual([x.type, x.id, x.commonWindow, x.maskPresent, x.hasPns], [e.type, e.id, e.commonWindow, e.maskPresent, e.hasPns]);
                e.ch.forEach(function (c, k) {
                    assert.deepStrictEqual([x.ch[k].windowSequence, x.ch[k].windowShape, x.ch[k].maxSFB, x.ch[k].groupLength], [c.windowSequence, c.windowShape, c.maxSFB, c.groupLength]);
                    assert.strictEqual(!!x.ch[k].tns, !!c.tns);
                    if (c.tns) { c.tns.short = c.windowSequence === 2; assert.deepStrictEqual(host.packTns([x.ch[k].tns]), host.packTns([c.tns]), 'frame ' + n + ': TNS side info'); }
                });
            });
        }
        n++;
    }
    assert.strictEqual(n, 41);
    const pcm = [];
    for (const Front of [host.FrontEnd, host.GpuFrontEnd]) {
        const dec = new host.GpuAACDecoder({ frontend: new Front({ codebooks: cb }), lookahead: 64 });
        dec.init(); dec.setCookie(new Uint8Array([(2 << 3) | (3 >> 1), ((3 & 1) << 7) | (2 << 3)]));
        dec.feed(new Uint8Array(stream));
        const out = [];
        for (let chunk; (chunk = dec.readChunk()) !== null;) out.push(chunk);
        assert.strictEqual(out.length, 40);
        pcm.push(out);
    }
    for (let t = 0; t < 40; t++) assert.deepStrictEqual(pcm[1][t], pcm[0][t], 'frame ' + t + ': PCM');
    console.log('GpuFrontEnd ok: 41 frames parsed on the device, 40 decoded');
}
(async function () {
    // decodeBatchAsync: the event loop stays free while the GPU decodes (a timer fires before the promise resolves or right after)
    const name = 'scn_stereo', ref = g[name + '.pcm'];
    const eng = new host.Engine({ maxStreams: 1, maxChannels: 2 });
    const pcm = new Float32Array(ref.length);
    const out = await eng.decodeBatchAsync(g[name + '.units'], g[name + '.q'], g[name + '.meta'], pcm);
    assert.strictEqual(out, pcm);
    assert.ok(rms(pcm, ref) < 1e-5);
    let rejected = false;
    try { await eng.decodeBatchAsync(g[name + '.units'], new Int16Array(1024), g[name + '.meta'], new Float32Array(2048)); } catch (e) { rejected = /aacgpu/.test(e.message); }
    assert.ok(rejected, 'async error path');
    console.log('decodeBatchAsync ok');
    console.log('host gpu tests ok');
})().catch(function (e) { console.error(e); process.exit(1); });
