#!/usr/bin/env node
/* The plugin under an Aurora stand-in (tests/js/av_stub.js):  node tests/js/test_aurora.js [cpu|gpu]
 *  cpu: registration ('mp4a', 'aac ', the ADTS demuxer's probe), the demuxer's events, and the whole chain
 *       source -> demuxer -> decoder with a recording engine: every frame of a committed stream reaches the engine once,
 *       in order, however the source cuts the bytes (Aurora rewinds the stream after a readChunk() that returned nothing)
 *  gpu: the same chain with the real engine: the PCM Aurora's 'data' events carry equals what the reference's own
 *       readChunk() produced from the same bytes (tests/golden/streams/*.refpcm) */
'use strict';
const fs = require('fs'), path = require('path'), assert = require('assert');
const root = path.join(__dirname, '..', '..');
const AV = require('./av_stub.js');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const mode = process.argv[2] || 'cpu';
const streams = path.join(root, 'tests', 'golden', 'streams');
const manifest = JSON.parse(fs.readFileSync(path.join(streams, 'manifest.json')));

function play(name, pieces, options) {
    AV.Demuxer.registry.length = 0;                              // one registration per process in real use; the test registers per run
    const plugin = require(path.join(root, 'aac.js_amd', 'js', 'aurora.js')).register(AV, options);
    const bytes = new Uint8Array(fs.readFileSync(path.join(streams, name + '.aac')));
    const source = new AV.EventEmitter(), out = { pcm: [], events: [], errors: [] };
    const first = new AV.Buffer(bytes.subarray(0, pieces[0]));
    const DemuxerClass = AV.Demuxer.find(first);
    assert.strictEqual(DemuxerClass, plugin.Demuxer, 'the ADTS demuxer answers the probe');
    let decoder = null;
    const demuxer = Object.create(DemuxerClass.prototype);
    /* Aurora's Asset: on 'format' it looks the decoder up by formatID and constructs it */
    AV.EventEmitter.call(demuxer);
    demuxer.on('format', function (format) {
        out.events.push('format');
        out.format = format;
        const DecoderClass = AV.Decoder.find(format.formatID);
        assert.strictEqual(DecoderClass, plugin.Decoder);
        decoder = new DecoderClass(demuxer, format);
        decoder.on('data', function (pcm) { out.pcm.push(pcm); });
        decoder.on('error', function (e) { out.errors.push(e); });
        decoder.on('end', function () { out.ended = true; });
    });
    demuxer.on('cookie', function () { out.events.push('cookie'); });
    DemuxerClass.call(demuxer, source, null);
    let at = 0, i = 0;
    while (at < bytes.length) {
        const n = Math.min(pieces[i % pieces.length], bytes.length - at);
        source.emit('data', new AV.Buffer(bytes.subarray(at, at + n)));
        at += n; i++;
        while (decoder && decoder.decode()) {}                    // Aurora's queue keeps calling decode() while it yields
    }
    source.emit('end');
    while (decoder && decoder.decode()) {}
    out.decoder = decoder;
    return out;
}

/* The 'mp4a' route: Aurora's M4A demuxer hands the decoder a track's samples without ADTS headers, and one buffer may hold
 * several contiguous samples (a chunk); the reference reads block after block from the same bitstream (decoder.js:129-199).
 * The ADTS test streams with their headers cut off are such samples; `groups` = samples per emitted buffer, cycled. */
function playMp4(name, groups, options) {
    AV.Demuxer.registry.length = 0;
    const plugin = require(path.join(root, 'aac.js_amd', 'js', 'aurora.js')).register(AV, options);
    const adts = require(path.join(root, 'aac.js_amd', 'js', 'adts.js'));
    const bytes = new Uint8Array(fs.readFileSync(path.join(streams, name + '.aac')));
    const list = adts.frames(bytes), h = list[0].header;
    const format = { formatID: 'mp4a', sampleRate: host.SAMPLE_RATES[h.samplingIndex], channelsPerFrame: h.chanConfig, bitsPerChannel: 16 };
    const demuxer = new AV.EventEmitter(), out = { pcm: [], errors: [] };
    const DecoderClass = AV.Decoder.find('mp4a');
    assert.strictEqual(DecoderClass, plugin.Decoder);
    const decoder = new DecoderClass(demuxer, format);
    decoder.on('data', function (pcm) { out.pcm.push(pcm); });
    decoder.on('error', function (e) { out.errors.push(e); });
    decoder.on('end', function () { out.ended = true; });
    demuxer.emit('cookie', new AV.Buffer(adts.cookie(h)));
    for (let i = 0, g = 0; i < list.length; g++) {
        const n = Math.min(groups[g % groups.length], list.length - i);
        let total = 0;
        for (let k = i; k < i + n; k++) total += list[k].length - list[k].header.headerBytes;
        const chunk = new Uint8Array(total);
        for (let k = i, at = 0; k < i + n; k++) {
            const block = bytes.subarray(list[k].offset + list[k].header.headerBytes, list[k].offset + list[k].length);
            chunk.set(block, at); at += block.length;
        }
        demuxer.emit('data', new AV.Buffer(chunk));
        i += n;
        while (decoder.decode()) {}
    }
    demuxer.emit('end');
    while (decoder.decode()) {}
    return out;
}

{
    AV.Demuxer.registry.length = 0;
    const plugin = require(path.join(root, 'aac.js_amd', 'js', 'aurora.js')).register(AV, { engine: function () { return { resetStream: function () {}, decodeBatch: function () {} }; } });
    assert.strictEqual(AV.Decoder.find('mp4a'), plugin.Decoder);
    assert.strictEqual(AV.Decoder.find('aac '), plugin.Decoder);
    assert.strictEqual(AV.Demuxer.find(new AV.Buffer(new Uint8Array([0, 1, 2, 3, 4, 5, 6, 7]))), null);
    assert.strictEqual(typeof plugin.Demuxer.readHeader, 'function');
}

if (mode === 'cpu') {
    /* the package's main: require('av'); require('aac.js_amd') registers as a side effect and exports the decoder class, like
     * the reference's package.json main (src/decoder.js) — `av` resolved to the stand-in through NODE_PATH */
    const os = require('os'), cp = require('child_process');
    const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'av-stub-'));
    fs.writeFileSync(path.join(dir, 'av.js'), 'module.exports = require(' + JSON.stringify(path.join(__dirname, 'av_stub.js')) + ');\n');
    const code = "const AV = require('av'); const D = require(" + JSON.stringify(path.join(root, 'aac.js_amd')) + ");" +
                 "if (AV.Decoder.find('mp4a') !== D || AV.Decoder.find('aac ') !== D || typeof D.Demuxer.readHeader !== 'function') process.exit(3);" +
                 "console.log('main ok ' + JSON.stringify(Object.keys(D.prototype).sort()));";
    const r = cp.spawnSync(process.execPath, ['-e', code], { env: Object.assign({}, process.env, { NODE_PATH: dir, AACG_OPTIONS: '{"lookahead":4}' }), encoding: 'utf8' });
    assert.strictEqual(r.status, 0, r.stdout + r.stderr);
    assert.ok(/main ok .*init.*readChunk.*setCookie/.test(r.stdout), r.stdout);
    fs.unlinkSync(path.join(dir, 'av.js')); fs.rmdirSync(dir);
}

if (mode === 'cpu') {
    /* players come and go: with options.shared every decoder Aurora constructs takes a stream slot, and gives it back when its
     * stream has ended and its last frame is out — five players one after the other on a SharedEngine of TWO slots */
    const engine = { resetStream: function () {}, decodeBatch: function (u, q, meta, pcm) { pcm.fill(1); } };
    const shared = new host.SharedEngine({ maxStreams: 2, maxChannels: 8, engine: function () { return engine; } });
    AV.Demuxer.registry.length = 0;
    require(path.join(root, 'aac.js_amd', 'js', 'aurora.js')).register(AV, { shared: shared, lookahead: 4 });
    const slots = [];
    for (let k = 0; k < 5; k++) {
        const out = play('stereo48', [4096], { shared: shared, lookahead: 4 });
        assert.strictEqual(out.errors.length, 0, String(out.errors[0]));
        assert.ok(out.ended && out.pcm.length === manifest.find(function (m) { return m.name === 'stereo48'; }).frames);
        slots.push(shared.groups.get(3).decoders.length);
    }
    assert.deepStrictEqual(slots, [0, 0, 0, 0, 0], 'a finished player still holds its slot');
}

for (const c of manifest) {
    if (mode === 'cpu') {
        /* a recording engine: what reaches aacg_decode_batch, frame by frame */
        for (const pieces of [[1 << 20], [997], [7, 300, 1], [64]]) {
            let units = 0;
            const engine = { resetStream: function () {}, decodeBatch: function (u, q, meta, pcm) { units += u.length / host.UNIT_BYTES; pcm.fill(1); } };
            const out = play(c.name, pieces, { engine: function () { return engine; }, lookahead: 8 });
            assert.deepStrictEqual(out.events.slice(0, 2), ['format', 'cookie']);
            assert.strictEqual(out.format.channelsPerFrame, c.channels);
            assert.strictEqual(out.errors.length, 0, String(out.errors[0]));
            assert.strictEqual(out.pcm.length, c.frames, c.name + ' pieces ' + pieces + ': frames delivered');
            assert.ok(out.pcm.every(function (p) { return p.length === 1024 * c.channels; }));
            const perFrame = fs.statSync(path.join(streams, c.name + '.units')).size / host.UNIT_BYTES / c.frames;
            assert.strictEqual(units, c.frames * perFrame, c.name + ': every element reached the engine exactly once');
            assert.ok(out.ended);
        }
        for (const groups of [[1], [3], [1, 5, 2], [1000]]) {       // samples per buffer, as an M4A demuxer may emit them
            let units = 0;
            const engine = { resetStream: function () {}, decodeBatch: function (u, q, meta, pcm) { units += u.length / host.UNIT_BYTES; pcm.fill(1); } };
            const out = playMp4(c.name, groups, { engine: function () { return engine; }, lookahead: 8 });
            assert.strictEqual(out.errors.length, 0, String(out.errors[0]));
            assert.strictEqual(out.pcm.length, c.frames, c.name + " 'mp4a' samples per buffer " + groups + ': frames delivered');
            const perFrame = fs.statSync(path.join(streams, c.name + '.units')).size / host.UNIT_BYTES / c.frames;
            assert.strictEqual(units, c.frames * perFrame, c.name + " 'mp4a': every element reached the engine exactly once");
            assert.ok(out.ended);
        }
    } else {
        {
            /* 'mp4a' with several samples per buffer through the real engine, JavaScript and device front ends */
            const refm = new Float32Array(new Uint8Array(fs.readFileSync(path.join(streams, c.name + '.refpcm'))).buffer);
            for (const opt of [{ lookahead: 16 }, { lookahead: 16, gpuParse: true }]) {
                const out = playMp4(c.name, [2, 7, 1], opt);
                assert.strictEqual(out.errors.length, 0, String(out.errors[0]));
                assert.strictEqual(out.pcm.length, c.frames, c.name + " 'mp4a' " + JSON.stringify(opt));
                let err = 0, n = 0;
                out.pcm.forEach(function (p, t) { for (let i = 0; i < p.length; i++, n++) { const d = p[i] - refm[t * p.length + i]; err += d * d; } });
                assert.ok(Math.sqrt(err / n) < 1e-5, c.name + " 'mp4a': rms " + Math.sqrt(err / n));
            }
        }
        const ref = new Float32Array(new Uint8Array(fs.readFileSync(path.join(streams, c.name + '.refpcm'))).buffer);
        for (const pieces of [[1 << 20], [1500, 13]]) {
            const out = play(c.name, pieces, { lookahead: 16 });
            assert.strictEqual(out.errors.length, 0, String(out.errors[0]));
            assert.strictEqual(out.pcm.length, c.frames);
            let err = 0, sig = 0, n = 0;
            out.pcm.forEach(function (p, t) { for (let i = 0; i < p.length; i++, n++) { const d = p[i] - ref[t * p.length + i]; err += d * d; sig += ref[t * p.length + i] * ref[t * p.length + i]; } });
            err = Math.sqrt(err / n); sig = Math.sqrt(sig / n);
            assert.ok(err < 1e-5 && err <= 5e-6 * sig, c.name + ': rms ' + err + ' of ' + sig);
            console.log(c.name + ' through av stub: ' + c.frames + ' frames, rms error ' + err.toExponential(2));
        }
    }
}
console.log('aurora ' + mode + ' tests ok');
