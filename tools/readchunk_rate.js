#!/usr/bin/env node
/* Plugin-level rate: frames per second out of readChunk(), bytes in -> PCM out, on one JavaScript thread.
 *   node tools/readchunk_rate.js [repeats]
 *   node tools/readchunk_rate.js --streams 256 [repeats]     N decoders on one SharedEngine (cross-stream batches), read round robin,
 *                                                            next to N decoders with an engine each: engine time per frame, checksum;
 *                                                            and on a RESIDENT SharedEngine (front end on the device too: one native
 *                                                            call per flush, bytes in, PCM out)
 * A long ADTS stream (the committed tests/golden/streams/stereo48.aac repeated) through
 *   - GpuAACDecoder with the JavaScript front end (parse on the CPU, transform on the GPU),
 *   - GpuAACDecoder with the device front end (parse and transform on the GPU),
 *   - the reference's own AACDecoder.readChunk() when the reference checkout is present (build container only; CPU).
 * The GPU lines need the built engine and a GPU; without one they are reported as null. */
'use strict';
const fs = require('fs'), path = require('path');
const root = path.join(__dirname, '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const argv0 = process.argv.slice(2);
/* --file NAME: another of the committed streams (tests/golden/streams/NAME.aac; surround48 = SCE + CPE + CPE + LFE, six channels) */
const fileAt = argv0.indexOf('--file'), fileName = fileAt >= 0 ? argv0[fileAt + 1] : 'stereo48';
const argv = argv0.filter(function (a, i) { return fileAt < 0 || (i !== fileAt && i !== fileAt + 1); });
const nStreams = argv[0] === '--streams' ? parseInt(argv[1], 10) : 0;
const repeats = parseInt((nStreams ? argv[2] : argv[0]) || (nStreams ? '4' : '400'), 10);
const one = new Uint8Array(fs.readFileSync(path.join(root, 'tests', 'golden', 'streams', fileName + '.aac')));
const nChannels = host.adts.frames(one)[0].header.chanConfig;
const perFile = host.adts.frames(one).length;
const bytes = new Uint8Array(one.length * repeats);
for (let i = 0; i < repeats; i++) bytes.set(one, i * one.length);
const out = { stream: fileName + '.aac x ' + repeats, channels: nChannels, frames: perFile * repeats, bytes: bytes.length };

function ours(gpuParse) {
    try {
        const fe = gpuParse ? new host.GpuFrontEnd({ batch: 4096 }) : new host.FrontEnd();
        const dec = new host.GpuAACDecoder({ frontend: fe, lookahead: 4096 });
        dec.init();
        const demux = new host.adts.AdtsDemuxer(function (event, payload) {
            if (event === 'format') Object.assign(dec.format, payload);
            else if (event === 'cookie') dec.setCookie(payload);
            else if (event === 'data') dec.feed(payload);
        });
        const t0 = process.hrtime.bigint();
        demux.push(bytes);
        let n = 0, sum = 0;
        for (let pcm; (pcm = dec.readChunk()); n++) sum += pcm[17];
        const s = Number(process.hrtime.bigint() - t0) / 1e9;
        return { frames_per_s: Math.round(n / s), frames: n, seconds: +s.toFixed(3), checksum: sum };
    } catch (e) { return { error: String(e.message || e).slice(0, 160) }; }
}
/* N concurrent streams: the same bytes into N decoders, read round robin as N players would; `shared`: one SharedEngine
 * (one batch per flush for all of them) or an engine per decoder (one batch per decoder: what N independent plugin instances
 * do).  Engine time = wall time inside engine.decodeBatch (upload, kernels, download), on this one JavaScript thread. */
/* yieldEvery > 0 (manyAsync): the same drain, but back to the event loop after every yieldEvery round-robin passes — what an
 * event-driven host (Aurora decodes from timers and 'data' events) does anyway, and what lets the garbage collector's finalizers
 * return a flush's page-locked PCM buffer to the addon's pool instead of a fresh 33 MB being page-locked per flush */
function many(shared, lookahead, pcmRing, overlap, yieldEvery, done) {
    const resident = shared === 'resident';
    try {
        let engineNs = 0n, batches = 0;
        const timed = function (eng) {
            const inner = eng.decodeBatch.bind(eng);
            eng.decodeBatch = function () { const t = process.hrtime.bigint(); try { return inner.apply(null, arguments); } finally { engineNs += process.hrtime.bigint() - t; batches++; } };
            return eng;
        };
        const sh = shared ? new host.SharedEngine({ maxStreams: nStreams, maxChannels: Math.max(2, nChannels), resident: resident, lookahead: lookahead, pcmRing: pcmRing | 0, overlap: !!overlap }) : null;
        const finish = function (t0, n, sum) {
            if (resident) { engineNs = sh.stats.engineNs; batches = sh.stats.batches; }       // wall time inside the one native call per flush
            const s = Number(process.hrtime.bigint() - t0) / 1e9, es = Number(engineNs) / 1e9;
            return { streams: nStreams, lookahead: lookahead, frames: n, seconds: +s.toFixed(3), frames_per_s: Math.round(n / s), batches: batches,
                     frames_per_batch: +(n / batches).toFixed(1), engine_seconds: +es.toFixed(3), frames_per_engine_second: Math.round(n / es), checksum: sum };
        };
        const decs = [];
        for (let i = 0; i < nStreams; i++) {
            const dec = new host.GpuAACDecoder({ frontend: resident ? null : new host.FrontEnd(), lookahead: lookahead, shared: sh });
            dec.init();
            const demux = new host.adts.AdtsDemuxer(function (event, payload) {
                if (event === 'format') Object.assign(dec.format, payload);
                else if (event === 'cookie') dec.setCookie(payload);
                else if (event === 'data') dec.feed(payload);
            });
            demux.push(bytes);
            if (!shared) timed(dec.engine);
            decs.push(dec);
        }
        if (shared && !resident) for (const g of sh.groups.values()) timed(g.engine);
        const t0 = process.hrtime.bigint();
        let n = 0, sum = 0, passes = 0;
        const pass = function () {
            let live = 0;
            for (const d of decs) { const pcm = d.readChunk(); if (pcm) { n++; sum += pcm[17]; live++; } }
            return live;
        };
        if (yieldEvery) {
            const turn = function () {
                try {
                    for (let k = 0; k < yieldEvery; k++) if (!pass()) return done(Object.assign(finish(t0, n, sum), { event_loop_turn_every_passes: yieldEvery }));
                    setImmediate(turn);
                } catch (e) { done({ error: String(e.message || e).slice(0, 200) }); }
            };
            return turn();
        }
        while (pass()) passes++;
        return finish(t0, n, sum);
    } catch (e) { const r = { error: String(e.message || e).slice(0, 200) }; if (done) done(r); return r; }
}
if (nStreams && process.env.READCHUNK_ONLY === 'event_loop') {
    many('resident', 16, process.env.READCHUNK_RING | 0, true, parseInt(process.env.READCHUNK_YIELD || '16', 10), function (r) { console.log(JSON.stringify(r)); process.exit(0); });
} else if (nStreams) {
    out.stream = fileName + '.aac x ' + repeats + ' x ' + nStreams + ' streams';
    out.independent_decoders = many(false, 16);
    out.shared_engine = many(true, 16);
    out.shared_engine_resident_no_overlap = many('resident', 16, 0, false);   // bytes -> PCM in one native call per flush, the flush waits for what it returns (round 5's default)
    out.shared_engine_resident = many('resident', 16, 0, true);   // the default since round 6: the next flush's batch is decoded while this one is read
    out.shared_engine_resident_ring8_no_overlap = many('resident', 16, 8, false);   // the PCM in a ring of 8 page-locked buffers (a frame stays valid for 7 more flushes)
    out.shared_engine_resident_ring8 = many('resident', 16, 8, true);
    /* (READCHUNK_ONLY=event_loop runs the default from a host that returns to the event loop every READCHUNK_YIELD passes, in a
     * process of its own: behind the synchronous drains above, the first turn of the event loop finalizes every page-locked
     * buffer those left behind — a few hundred hipHostFree calls that have nothing to do with the variant being timed) */
    const same = function (k) { return out[k] && out[k].checksum === out.independent_decoders.checksum; };
    if (out.shared_engine_resident.checksum !== undefined)
        out.resident_same_checksum = ['shared_engine_resident_no_overlap', 'shared_engine_resident', 'shared_engine_resident_ring8_no_overlap', 'shared_engine_resident_ring8'].every(same);
    if (out.independent_decoders.frames_per_engine_second && out.shared_engine.frames_per_engine_second) {
        out.engine_time_ratio = +(out.shared_engine.frames_per_engine_second / out.independent_decoders.frames_per_engine_second).toFixed(2);
        out.same_checksum = out.shared_engine.checksum === out.independent_decoders.checksum;
    }
    console.log(JSON.stringify(out, null, 1));
    process.exit(0);
}
out.gpu_transform_js_parse = ours(false);
out.gpu_transform_gpu_parse = ours(true);

const REF = '/root/reference/src/';
if (fs.existsSync(REF + 'decoder.js')) {
    process.env.NODE_PATH = path.join(root, 'tests', 'golden', 'gen', 'stubs');
    require('module').Module._initPaths();
    const { BitStream } = require(path.join(root, 'aac.js_amd', 'js', 'bits.js'));
    const AACDecoder = require(REF + 'decoder.js');
    const FilterBank = require(REF + 'filter_bank.js');
    const ref = new AACDecoder(); ref.format = {};
    ref.config = { profile: 2, sampleIndex: 3, sampleRate: 48000, chanConfig: 2, frameLength: 1024 };   // what setCookie leaves (decoder.js:53-113)
    ref.filter_bank = new FilterBank(false, 2);                                                         // decoder.js:112
    const frames = host.adts.frames(one).map(function (f) { return one.subarray(f.offset + (f.headerLength || 7), f.offset + f.length); });
    const t0 = process.hrtime.bigint();
    let n = 0;
    const budget = 5e9;
    while (Number(process.hrtime.bigint() - t0) < budget) { for (const f of frames) { ref.bitstream = new BitStream(f); ref.readChunk(); } n += frames.length; }
    out.reference_readChunk = { frames_per_s: Math.round(n / (Number(process.hrtime.bigint() - t0) / 1e9)), frames: n };
}
console.log(JSON.stringify(out, null, 1));
