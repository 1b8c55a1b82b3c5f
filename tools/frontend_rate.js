#!/usr/bin/env node
/* Parse rate of the JavaScript front end (frames/s on one core), next to the reference's own parser on the same
 * bytes when the reference checkout is present (its readChunk() with process() stubbed out, i.e. element loop +
 * Huffman + its in-loop dequantisation).  Streams: the committed tests/golden/streams/*.aac.
 *   node tools/frontend_rate.js [seconds] */
'use strict';
const fs = require('fs'), path = require('path');
const root = path.join(__dirname, '..');
const { FrontEnd } = require(path.join(root, 'aac.js_amd', 'js', 'frontend.js'));
const codebooks = require(path.join(root, 'aac.js_amd', 'js', 'codebooks.js'));
const adts = require(path.join(root, 'aac.js_amd', 'js', 'adts.js'));
const { BitStream } = require(path.join(root, 'aac.js_amd', 'js', 'bits.js'));
const REF = '/root/reference/src/', haveRef = fs.existsSync(REF + 'decoder.js');
const seconds = parseFloat(process.argv[2] || '2');
const cb = codebooks.standard();
let ref = null;
if (haveRef) {
    process.env.NODE_PATH = path.join(root, 'tests', 'golden', 'gen', 'stubs');
    require('module').Module._initPaths();
    const AACDecoder = require(REF + 'decoder.js');
    ref = new AACDecoder(); ref.format = {}; ref.process = function () { this.data = []; };
}
const out = {};
for (const c of JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'streams', 'manifest.json')))) {
    const bytes = new Uint8Array(fs.readFileSync(path.join(root, 'tests', 'golden', 'streams', c.name + '.aac')));
    const frames = adts.frames(bytes).map(function (f) { return bytes.subarray(f.offset, f.offset + f.length); });
    const config = { profile: 2, sampleIndex: c.sampleIndex, chanConfig: c.channels, frameLength: 1024 };
    const fe = new FrontEnd({ codebooks: cb });
    function time(one) {
        let n = 0;
        const t0 = process.hrtime.bigint();
        while (Number(process.hrtime.bigint() - t0) < seconds * 1e9) { for (const f of frames) one(f); n += frames.length; }
        return n / (Number(process.hrtime.bigint() - t0) / 1e9);
    }
    const mine = time(function (f) { fe.pushPacket(f); fe.parseFrame({ config: config }); });
    const r = { channels: c.channels, bytes_per_frame: Math.round(bytes.length / frames.length), frontend_frames_per_s: Math.round(mine) };
    if (ref) { ref.config = config; r.reference_parser_frames_per_s = Math.round(time(function (f) { ref.bitstream = new BitStream(f); ref.readChunk(); })); }
    out[c.name] = r;
}
console.log(JSON.stringify(out, null, 1));
