#!/usr/bin/env node
/* Coupling channel elements through the JavaScript host (CCE_SPEC):  node tests/js/test_coupling.js <outdir> [cpu|gpu]
 *
 * Streams written by the synthetic writer (aac_writer.js, the standard's syntax) with coupling elements whose targets are
 * the frame's real elements.  Checked here: what FrontEnd({ coupling: true, referenceQuirks: false }) returns for them —
 * coupling point, targets, and every gain, recomputed in this file from what the writer was told to write (cce.js:77-107's
 * arithmetic) — and that GpuAACDecoder({ cceMode: CCE_SPEC }) turns them into units + aacg_cce_info records with the right
 * channels.  The records the decoder hands to the engine go to <outdir> for tests/test_cce_spec.py (oracle / emulator /
 * GPU); in gpu mode the PCM the real engine returned through readChunk() goes there too. */
'use strict';
const fs = require('fs'), path = require('path'), assert = require('assert');
const root = path.join(__dirname, '..', '..');
const host = require(path.join(root, 'aac.js_amd', 'js'));
const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
const { FrontEnd } = require(path.join(root, 'aac.js_amd', 'js', 'frontend.js'));
const { Writer, Rng } = require('./aac_writer.js');
const { randomFrame, PATTERN } = require('./stream_cases.js');
const outdir = process.argv[2], mode = process.argv[3] || 'cpu';
if (!outdir) { console.error('usage: test_coupling.js <outdir> [cpu|gpu]'); process.exit(2); }
fs.mkdirSync(outdir, { recursive: true });

const cb = codebooks.standard(), SI = 3, wr = new Writer(cb, SI), rng = new Rng(0xCCE5);
const layout = ['cce', 'cpe', 'sce', 'cce'], C = 3, FRAMES = 12;
const SCALE = [Math.pow(2, 1 / 8), Math.pow(2, 1 / 4), Math.SQRT2, 2];
const written = [], frames = [];
for (let t = 0; t < FRAMES; t++) {
    const elements = randomFrame(wr, rng, layout, function () { return PATTERN[t % PATTERN.length]; }, { quirks: false, cceNoZero: t % 2 === 0 });
    const cpe = elements[1], sce = elements[2];
    cpe.id = 1; sce.id = 4;
    elements.forEach(function (e, i) {
        if (e.type !== 'cce') return;
        /* as coded: ind_sw_cce_flag << 1 | cc_domain.  The first coupling element of every frame is independently switched
         * (it owns overlap state, so it has to be there in every frame of a batch: include/aacgpu.h), the second is dependent */
        e.point = i === 0 ? 2 + (t & 1) : (t >> 1) & 1;
        /* targets that exist: the pair (one channel, the other, both with one list, both with two), the single channel,
         * and one that matches nothing (skipped, but its gain lists are still in the stream) */
        const sel = (t + i) % 4;
        e.targets = [{ pair: true, id: 1, sel: sel }, { pair: false, id: 9, sel: 2 }, { pair: false, id: 4, sel: 2 }];
        const n = e.targets.reduce(function (a, x) { return a + 1 + (x.pair && x.sel === 3 ? 1 : 0); }, 0);
        e.lists = [];
        for (let l = 1; l < n; l++) e.lists.push({ cge: rng.below(2), common: rng.below(21) - 10, steps: [rng.below(9) - 4, rng.below(9) - 4, 0] });
    });
    written.push(elements);
    frames.push(wr.adtsFrame(elements, C, { crc: t % 3 === 0 }));
}

/* what the gains must be, from the writer's element (cce.js:77-107; gains by band) */
function expectedGains(e) {
    const independent = (e.point & 2) !== 0, ch = e.ch[0], nb = ch.groupLen.length * ch.maxSFB, scale = SCALE[e.scale], out = [];
    const all = [{ cge: 1, common: 0, first: true }].concat(e.lists);
    all.forEach(function (l) {
        const g = new Float32Array(120);
        let gain = l.first ? 0 : (independent || l.cge ? l.common : 0), cache = Math.pow(scale, -gain), s = 0;
        if (independent) g[0] = cache;
        else for (let b = 0; b < nb; b++) {
            if (ch.bandTypes[b] === 0) continue;
            if (!l.first && !l.cge) {
                let t = l.steps[s++ % l.steps.length];
                if (t !== 0) {
                    let sg = 1;
                    t = gain += t;
                    if (!e.sign) { sg -= 2 * (t & 1); t >>>= 1; }
                    cache = Math.pow(scale, -t) * sg;
                }
            }
            g[b] = cache;
        }
        out.push(g);
    });
    return out;
}

/* 1. the front end's view of the coupling elements */
{
    const fe = new FrontEnd({ codebooks: cb, coupling: true, referenceQuirks: false });
    for (const f of frames) fe.push(f);
    for (let t = 0; t < FRAMES; t++) {
        const frame = fe.parseFrame({ config: { sampleIndex: SI } });
        assert.deepStrictEqual(frame.elements.map(function (e) { return e.type; }), ['cce', 'cpe', 'sce', 'cce']);
        assert.strictEqual(frame.q.length, 5 * 1024);
        frame.elements.forEach(function (e, i) {
            if (e.type !== 'cce') return;
            const w = written[t][i];
            assert.strictEqual(e.couplingPoint, w.point & 2 ? 2 : w.point & 1, 'coupling point');
            assert.deepStrictEqual(e.targets, w.targets.map(function (x) { return { pair: x.pair, id: x.id, sel: x.pair ? x.sel : 2 }; }));
            const want = expectedGains(w);
            assert.strictEqual(e.gains.length, want.length, 'gain lists');
            e.gains.forEach(function (g, l) { assert.deepStrictEqual(Array.from(g), Array.from(want[l]), 'frame ' + t + ' element ' + i + ' list ' + l); });
        });
    }
    assert.strictEqual(fe.parseFrame({ config: { sampleIndex: SI } }), null);
    /* the default front end drops them, as the reference does */
    const plain = new FrontEnd({ codebooks: cb, referenceQuirks: false });
    plain.push(frames[0]);
    assert.deepStrictEqual(plain.parseFrame({ config: { sampleIndex: SI } }).elements.map(function (e) { return e.type; }), ['cpe', 'sce']);
}

/* 2. through the decoder: units + aacg_cce_info records */
const captured = { units: [], q: [], meta: [], cce: [] };
const recorder = { resetStream: function () {}, decodeBatch: function (units, q, meta, pcm, tns, cce) {
    captured.units.push(Buffer.from(units)); captured.q.push(Buffer.from(q.buffer, q.byteOffset, q.byteLength));
    captured.meta.push(Buffer.from(meta.buffer, meta.byteOffset, meta.byteLength)); captured.cce.push(Buffer.from(cce)); } };
function decode(engine, frontend) {
    const dec = new host.GpuAACDecoder({ engine: engine, frontend: frontend || new FrontEnd({ codebooks: cb, coupling: true, referenceQuirks: false }),
                                         lookahead: FRAMES, cceMode: host.CCE_SPEC, maxCoupling: 2 });
    dec.init();
    dec.setCookie(new Uint8Array([(2 << 3) | (SI >> 1), ((SI & 1) << 7) | (C << 3)]));
    for (const f of frames) dec.feed(f);
    const out = [];
    for (let pcm; (pcm = dec.readChunk()) !== null;) out.push(pcm);
    return out;
}
assert.strictEqual(decode(recorder).length, FRAMES);
assert.strictEqual(captured.units.length, 1);                   // one batch: look-ahead covers the stream
{
    const units = host.unpackUnits(new Uint8Array(captured.units[0])), view = new DataView(captured.cce[0].buffer, captured.cce[0].byteOffset);
    assert.strictEqual(units.length, FRAMES * 4);
    for (let t = 0; t < FRAMES; t++) {
        const u = units.slice(4 * t, 4 * t + 4);                  // cpe, sce, then the two coupling elements
        assert.deepStrictEqual(u.map(function (x) { return x.channel; }), [0, 2, 3, 4]);
        assert.deepStrictEqual(u.map(function (x) { return x.coefOffset - 5 * t; }), [1, 3, 0, 4]);       // blocks in bitstream order
        [2, 3].forEach(function (k, which) {
            const rec = new DataView(captured.units[0].buffer, captured.units[0].byteOffset + 64 * (4 * t + k)).getUint32(60, true);   // reserved1
            const o = host.CCE_BYTES * rec, w = written[t][which === 0 ? 0 : 3], sel = w.targets[0].sel;
            assert.strictEqual(view.getUint8(o), w.point & 2 ? 2 : w.point & 1);
            /* decoder.js:411-431 with a comparison that can match: pair target first (by its select), the unknown element skipped
             * (its list index still advances), then the single channel */
            const want = [];
            let index = 0;
            if (sel !== 1) { want.push([0, index]); if (sel) index++; }
            if (sel !== 2) want.push([1, index++]);
            index += 1;
            want.push([2, index]);
            assert.strictEqual(view.getUint8(o + 1), want.length);
            want.forEach(function (x, j) { assert.deepStrictEqual([view.getUint8(o + 4 + 2 * j), view.getUint8(o + 5 + 2 * j)], x, 'frame ' + t + ' target ' + j); });
        });
    }
}
fs.writeFileSync(path.join(outdir, 'coupling.units'), captured.units[0]);
fs.writeFileSync(path.join(outdir, 'coupling.q'), captured.q[0]);
fs.writeFileSync(path.join(outdir, 'coupling.meta'), captured.meta[0]);
fs.writeFileSync(path.join(outdir, 'coupling.cce'), captured.cce[0]);
fs.writeFileSync(path.join(outdir, 'coupling.json'), JSON.stringify({ channels: C, hidden: 2, frames: FRAMES, sampleIndex: SI }));
if (mode === 'gpu') {
    const pcm = decode(null);                                     // a real engine, created by setCookie
    assert.strictEqual(pcm.length, FRAMES);
    fs.writeFileSync(path.join(outdir, 'coupling.pcm'), Buffer.concat(pcm.map(function (p) { return Buffer.from(p.buffer, p.byteOffset, p.byteLength); })));
    /* the same through the device front end (round 6): a frame that holds a coupling element is flagged by the device parser and
     * its records come from the JavaScript front end; bit for bit the same PCM.  Every frame of this stream has coupling elements;
     * a stream without them stays on the device */
    const gfe = new host.GpuFrontEnd({ codebooks: cb, coupling: true, referenceQuirks: false });
    const viaDevice = decode(null, gfe);
    assert.strictEqual(viaDevice.length, FRAMES);
    viaDevice.forEach(function (p, t) { assert.deepStrictEqual(Buffer.from(p.buffer, p.byteOffset, p.byteLength), Buffer.from(pcm[t].buffer, pcm[t].byteOffset, pcm[t].byteLength), 'frame ' + t); });
    assert.deepStrictEqual(gfe.stats, { deviceFrames: 0, cpuFrames: FRAMES });
    const plainFrames = [];
    for (let t = 0; t < 4; t++) plainFrames.push(wr.adtsFrame(randomFrame(wr, rng, ['cpe', 'sce'], function () { return PATTERN[t % PATTERN.length]; }, { quirks: false }), C, {}));
    const gfe2 = new host.GpuFrontEnd({ codebooks: cb, coupling: true, referenceQuirks: false }), ref2 = new FrontEnd({ codebooks: cb, coupling: true, referenceQuirks: false });
    for (const f of plainFrames) { gfe2.push(f); ref2.push(f); }
    for (let t = 0; t < 4; t++) {
        const a = gfe2.parseFrame({ config: { sampleIndex: SI } }), b = ref2.parseFrame({ config: { sampleIndex: SI } });
        assert.deepStrictEqual(Array.from(a.q), Array.from(b.q)); assert.deepStrictEqual(Array.from(a.meta), Array.from(b.meta));
    }
    assert.deepStrictEqual(gfe2.stats, { deviceFrames: 4, cpuFrames: 0 });
}
console.log('coupling ' + mode + ' tests ok');
