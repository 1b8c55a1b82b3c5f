#!/usr/bin/env node
/* SharedEngine (aac.js_amd/js/shared_engine.js): N decoders, one engine per sample rate, ONE batch per flush.
 *   node tests/js/test_shared.js cpu   a recording engine: batch shapes, stream slots, order, error isolation (no GPU)
 *   node tests/js/test_shared.js gpu   8 interleaved streams through the real engine against the PCM the reference decoded from
 *                                      the same bytes (tests/golden/streams/*.refpcm), and against 8 independent decoders */
'use strict';
const fs = require('fs'), path = require('path'), assert = require('assert');
const root = path.join(__dirname, '..', '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const streams = path.join(root, 'tests', 'golden', 'streams');
const manifest = JSON.parse(fs.readFileSync(path.join(streams, 'manifest.json')));
const mode = process.argv[2] || 'cpu';

function open(name, opts) {
    const dec = new host.GpuAACDecoder(Object.assign({ frontend: new host.FrontEnd(), lookahead: 4 }, opts));
    dec.init();
    const demux = new host.adts.AdtsDemuxer(function (event, payload) {
        if (event === 'format') Object.assign(dec.format, payload);
        else if (event === 'cookie') dec.setCookie(payload);
        else if (event === 'data') dec.feed(payload);
    });
    demux.push(new Uint8Array(fs.readFileSync(path.join(streams, name + '.aac'))));
    return dec;
}
function drainRoundRobin(decs) {
    const out = decs.map(function () { return []; });
    for (let live = decs.length; live;) {
        live = 0;
        decs.forEach(function (d, i) { const pcm = d.readChunk(); if (pcm) { out[i].push(pcm); live++; } });
    }
    return out;
}
const names = ['stereo48', 'surround48', 'mono22', 'stereo48', 'extras8k', 'cce96', 'surround48', 'stereo48'];

if (mode === 'cpu') {
    /* recording engines, one per sample rate: what reaches decodeBatch */
    const calls = [];
    let poison = -1, poisonCode = -5;                  // a stream slot (of the 48 kHz engine) whose units the engine refuses, and with which status (include/aacgpu.h)
    const mk = function (maxStreams) {
        return new host.SharedEngine({ maxStreams: maxStreams, engine: function (sampleIndex) {
            return { resetStream: function () {}, decodeBatch: function (u, q, meta, pcm) {
                const units = host.unpackUnits(u);
                if (sampleIndex === 3 && units.some(function (x) { return x.stream === poison; })) throw new Error('aacgpu: aacg_decode_batch failed (' + poisonCode + '): engine refuses stream ' + poison);
                calls.push({ sampleIndex: sampleIndex, units: units });
                for (const x of units) for (let c = 0; c < x.ch.length; c++) pcm[x.pcmOffset + x.channel + c] = 1000 * x.stream + 1;   // sample 0 of every channel: its stream
            } };
        } });
    };
    const shared = mk(8);
    const decs = names.map(function (n) { return open(n, { shared: shared }); });
    /* slots: dense from 0 per sample rate (one engine each: 48 kHz, 22.05 kHz, 8 kHz, 96 kHz) */
    assert.deepStrictEqual(decs.map(function (d) { return d.stream; }), [0, 1, 0, 2, 0, 0, 3, 4]);
    assert.strictEqual(shared.groups.size, 4);
    const got = drainRoundRobin(decs);
    names.forEach(function (n, i) {
        const c = manifest.find(function (m) { return m.name === n; });
        assert.strictEqual(got[i].length, c.frames, n + ': frames delivered');
        for (const p of got[i]) {
            assert.strictEqual(p.length, 1024 * c.channels);
            for (let ch = 0; ch < c.channels; ch++) assert.strictEqual(p[ch], 1000 * decs[i].stream + 1, n + ': a frame of another stream');
        }
    });
    /* the first 48 kHz batch holds the look-ahead (4 frames) of all five 48 kHz decoders */
    const first = calls.find(function (c) { return c.sampleIndex === 3; });
    assert.deepStrictEqual(Array.from(new Set(first.units.map(function (u) { return u.stream; }))).sort(), [0, 1, 2, 3, 4]);
    const blocks = first.units.map(function (u) { return u.coefOffset; });
    assert.strictEqual(new Set(blocks).size, blocks.length, 'two units share a coefficient block');
    assert.ok(shared.stats.batches < 30 && shared.stats.frames === names.reduce(function (a, n) { return a + manifest.find(function (m) { return m.name === n; }).frames; }, 0));

    /* an engine error on a shared batch costs the stream that caused it, nobody else */
    calls.length = 0; poison = 1;
    const sh2 = mk(8);
    const d2 = ['stereo48', 'stereo48', 'stereo48'].map(function (n) { return open(n, { shared: sh2 }); });
    assert.ok(d2[0].readChunk() instanceof Float32Array);
    assert.throws(function () { d2[1].readChunk(); }, /engine refuses stream 1/);
    assert.ok(d2[2].readChunk() instanceof Float32Array);
    assert.ok(sh2.stats.retries >= 1);
    /* ... but only an error raised BEFORE anything was launched: after a device error (AACG_ERR_NO_DEVICE) the launch may have
     * advanced every stream's state, decoding the same frames again would advance it twice — the error goes to every stream of the batch */
    poisonCode = -2;
    const sh4 = mk(8);
    const d4 = ['stereo48', 'stereo48', 'stereo48'].map(function (n) { return open(n, { shared: sh4 }); });
    for (const d of d4) assert.throws(function () { d.readChunk(); }, /failed \(-2\)/);
    assert.strictEqual(sh4.stats.retries, 0);
    poison = -1; poisonCode = -5;

    /* capacity and slot reuse */
    const sh3 = mk(1);
    const a = open('stereo48', { shared: sh3 });
    assert.throws(function () { open('stereo48', { shared: sh3 }); }, /more than 1 streams/);
    a.close();
    assert.strictEqual(open('stereo48', { shared: sh3 }).stream, 0);
    assert.throws(function () { open('stereo48', { shared: mk(4), tnsMode: host.TNS_SPEC }); }, /differ from the shared engine/);
} else if (mode === 'gpu') {
    const shared = new host.SharedEngine({ maxStreams: 16, maxChannels: 8 });
    const decs = names.map(function (n) { return open(n, { shared: shared }); });
    const got = drainRoundRobin(decs);
    const alone = drainRoundRobin(names.map(function (n) { return open(n, {}); }));
    names.forEach(function (n, i) {
        const c = manifest.find(function (m) { return m.name === n; });
        const ref = new Float32Array(new Uint8Array(fs.readFileSync(path.join(streams, n + '.refpcm'))).buffer);
        assert.strictEqual(got[i].length, c.frames, n + ': frames delivered');
        let err = 0, sig = 0, k = 0;
        got[i].forEach(function (p, t) {
            assert.strictEqual(p.length, 1024 * c.channels);
            for (let j = 0; j < p.length; j++, k++) { const d = p[j] - ref[t * p.length + j]; err += d * d; sig += ref[t * p.length + j] * ref[t * p.length + j]; }
            /* the same bits as a decoder with an engine of its own: streams only meet in the launch */
            assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(alone[i][t].buffer, alone[i][t].byteOffset, alone[i][t].byteLength), n + ' frame ' + t);
        });
        err = Math.sqrt(err / k); sig = Math.sqrt(sig / k);
        assert.ok(err < 1e-5 && err <= 5e-6 * sig, n + ': rms ' + err + ' of ' + sig);
    });
    /* 8 streams, look-ahead 4: a flush carries the buffered frames of every decoder of a sample rate, so the three stereo48 (18
     * frames each) and two surround48 (5 frames) decoders at 48 kHz share their batches */
    assert.ok(shared.stats.batches < 8 * 5, 'batches ' + shared.stats.batches);
    assert.strictEqual(shared.stats.frames, names.reduce(function (a, n) { return a + manifest.find(function (m) { return m.name === n; }).frames; }, 0));
    assert.strictEqual(shared.stats.retries, 0);
    console.log('shared engine: ' + shared.stats.frames + ' frames of 8 streams in ' + shared.stats.batches + ' batches (' + shared.groups.size + ' engines)');

    /* RESIDENT: the front end on the device too — one native call per flush, bytes in, PCM out.  The same eight streams, all of
     * them on the resident route now (one pipeline per sample rate and channel count; the 5.1 streams' element layout is learnt
     * on the device from their first frame, the coupling elements of cce96 are parsed and dropped as the reference executes
     * them); every stream against the PCM the reference decoded from the same bytes, and bit for bit against a decoder of its own */
    const res = new host.SharedEngine({ maxStreams: 16, maxChannels: 8, resident: true, lookahead: 4 });
    const rdecs = names.map(function (n) { return open(n, { shared: res }); });
    assert.deepStrictEqual(rdecs.map(function (d) { return !!d.resident; }), names.map(function () { return true; }));
    /* an 'mp4a' decoder on the same resident SharedEngine (ADVICE round 5: it used to be sent to the resident route, which reads
     * ADTS frame lengths, and delivered nothing): MP4 samples — the ADTS frames without their headers, three to a buffer — take
     * the parsing route and come out as the same PCM */
    {
        const bytes = new Uint8Array(fs.readFileSync(path.join(streams, 'stereo48.aac')));
        const list = host.adts.frames(bytes), h = list[0].header;
        const mp4 = new host.GpuAACDecoder({ frontend: new host.FrontEnd(), lookahead: 4, shared: res, format: { formatID: 'mp4a' } });
        mp4.init(); mp4.setCookie(host.adts.cookie(h));
        assert.ok(!mp4.resident, "an 'mp4a' decoder does not take the resident route");
        const out = [];
        for (let i = 0; i < list.length; i += 3) {
            const blocks = list.slice(i, i + 3).map(function (f) { return bytes.subarray(f.offset + f.header.headerBytes, f.offset + f.length); });
            const chunk = new Uint8Array(blocks.reduce(function (a, b) { return a + b.length; }, 0));
            let at = 0; for (const b of blocks) { chunk.set(b, at); at += b.length; }
            mp4.feedPacket(chunk, true);
            for (let p; (p = mp4.readChunk());) out.push(p);
        }
        assert.strictEqual(out.length, list.length, "'mp4a' on a resident SharedEngine: frames delivered");
        out.forEach(function (p, t) { assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(alone[0][t].buffer, alone[0][t].byteOffset, alone[0][t].byteLength), "'mp4a' frame " + t); });
        mp4.close();
    }
    const rgot = drainRoundRobin(rdecs);
    names.forEach(function (n, i) {
        const c = manifest.find(function (m) { return m.name === n; });
        assert.strictEqual(rgot[i].length, c.frames, n + ' (resident): frames delivered');
        rgot[i].forEach(function (p, t) {
            assert.strictEqual(p.length, 1024 * c.channels);
            assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(alone[i][t].buffer, alone[i][t].byteOffset, alone[i][t].byteLength), n + ' (resident) frame ' + t);
        });
    });
    /* a stream fed in pieces (frames split across feeds), and one with a corrupt frame in the middle: the error is thrown where the
     * frame is reached, the frames before it are delivered, the stream goes on */
    const whole = new Uint8Array(fs.readFileSync(path.join(streams, 'stereo48.aac')));
    const r2 = new host.SharedEngine({ maxStreams: 4, maxChannels: 2, resident: true, lookahead: 3 });
    const piecewise = new host.GpuAACDecoder({ shared: r2 }); piecewise.init();
    const demux = new host.adts.AdtsDemuxer(function (event, payload) {
        if (event === 'format') Object.assign(piecewise.format, payload); else if (event === 'cookie') piecewise.setCookie(payload); else if (event === 'data') piecewise.feed(payload);
    });
    const got2 = [];
    for (let at = 0; at < whole.length; at += 777) {
        demux.push(whole.subarray(at, Math.min(whole.length, at + 777)));
        for (let p; (p = piecewise.readChunk());) got2.push(p);
    }
    const ref0 = alone[0];
    assert.strictEqual(got2.length, ref0.length, 'piecewise feed: frames delivered');
    got2.forEach(function (p, t) { assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(ref0[t].buffer, ref0[t].byteOffset, ref0[t].byteLength), 'piecewise frame ' + t); });
    const list = host.adts.frames(whole), bad = new Uint8Array(whole);
    for (let k = 9; k < list[5].length; k++) bad[list[5].offset + k] = 0;     // frame 5: zeros behind the header (element after element until the frame runs out)
    const broken = new host.GpuAACDecoder({ shared: r2 }); broken.init();
    const dm2 = new host.adts.AdtsDemuxer(function (event, payload) {
        if (event === 'format') Object.assign(broken.format, payload); else if (event === 'cookie') broken.setCookie(payload); else if (event === 'data') broken.feed(payload);
    });
    dm2.push(bad);
    let threw = 0, delivered = 0;
    for (let t = 0; t < list.length; t++) {
        try { const p = broken.readChunk(); if (p) { delivered++; if (t < 5) assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(ref0[t].buffer, ref0[t].byteOffset, ref0[t].byteLength)); } }
        catch (e) { threw++; assert.strictEqual(t, 5, 'the error belongs to frame 5: ' + e.message); }
    }
    assert.strictEqual(threw, 1); assert.strictEqual(delivered, list.length - 1);
    /* { pcmRing: K }: the flushes' PCM in K page-locked buffers used in turn — a frame is valid until K more flushes: copied as it
     * comes (what a consumer that keeps frames does), it is the same PCM; held on to, it is overwritten by the K-th flush after it */
    const r3 = new host.SharedEngine({ maxStreams: 4, maxChannels: 2, resident: true, lookahead: 2, pcmRing: 3 });
    const ringDec = open('stereo48', { shared: r3 });
    const copies = [], views = [];
    for (let p; (p = ringDec.readChunk());) { copies.push(p.slice()); views.push(p); }
    assert.strictEqual(copies.length, ref0.length);
    copies.forEach(function (p, t) { assert.deepStrictEqual(Buffer.from(p.buffer), Buffer.from(ref0[t].buffer, ref0[t].byteOffset, ref0[t].byteLength), 'ring frame ' + t); });
    assert.notDeepStrictEqual(Buffer.from(views[0].buffer, views[0].byteOffset, views[0].byteLength), Buffer.from(copies[0].buffer), 'frame 0 of flush 0 was not overwritten by flush 3');
    /* { overlap: true }: the next flush's batch is decoded (a native thread) while the caller reads this one.  The same eight
     * streams round robin, a stream fed in pieces, and the corrupt frame: the same frames, the same bits, the error where it belongs */
    const ro = new host.SharedEngine({ maxStreams: 16, maxChannels: 8, resident: true, lookahead: 4, overlap: true });
    const odecs = names.map(function (n) { return open(n, { shared: ro }); });
    const ogot = drainRoundRobin(odecs);
    names.forEach(function (n, i) {
        const c = manifest.find(function (m) { return m.name === n; });
        assert.strictEqual(ogot[i].length, c.frames, n + ' (resident, overlapped): frames delivered');
        ogot[i].forEach(function (p, t) {
            assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(alone[i][t].buffer, alone[i][t].byteOffset, alone[i][t].byteLength), n + ' (resident, overlapped) frame ' + t);
        });
    });
    const r4 = new host.SharedEngine({ maxStreams: 4, maxChannels: 2, resident: true, lookahead: 3, overlap: true });
    const pw = new host.GpuAACDecoder({ shared: r4 }); pw.init();
    const dm4 = new host.adts.AdtsDemuxer(function (event, payload) {
        if (event === 'format') Object.assign(pw.format, payload); else if (event === 'cookie') pw.setCookie(payload); else if (event === 'data') pw.feed(payload);
    });
    const got4 = [];
    for (let at = 0; at < whole.length; at += 1501) {
        dm4.push(whole.subarray(at, Math.min(whole.length, at + 1501)));
        for (let p; (p = pw.readChunk());) got4.push(p);
    }
    assert.strictEqual(got4.length, ref0.length, 'piecewise feed, overlapped: frames delivered');
    got4.forEach(function (p, t) { assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(ref0[t].buffer, ref0[t].byteOffset, ref0[t].byteLength), 'piecewise overlapped frame ' + t); });
    const broken4 = new host.GpuAACDecoder({ shared: r4 }); broken4.init();
    const dm5 = new host.adts.AdtsDemuxer(function (event, payload) {
        if (event === 'format') Object.assign(broken4.format, payload); else if (event === 'cookie') broken4.setCookie(payload); else if (event === 'data') broken4.feed(payload);
    });
    dm5.push(bad);
    let threw4 = 0, delivered4 = 0;
    for (let t = 0; t < list.length; t++) {
        try { const p = broken4.readChunk(); if (p) { delivered4++; if (t < 5) assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(ref0[t].buffer, ref0[t].byteOffset, ref0[t].byteLength)); } }
        catch (e) { threw4++; assert.strictEqual(t, 5, 'the error belongs to frame 5: ' + e.message); }
    }
    assert.strictEqual(threw4, 1); assert.strictEqual(delivered4, list.length - 1);
    console.log('resident shared engine: ' + res.stats.frames + ' frames in ' + res.stats.batches + ' native calls; piecewise feed and a corrupt frame ok; overlapped flushes ok (' + ro.stats.batches + ' native calls)');
}
console.log('shared ' + mode + ' tests ok');
