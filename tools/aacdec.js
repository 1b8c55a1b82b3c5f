#!/usr/bin/env node
/* Decode an ADTS .aac file on the GPU:  node tools/aacdec.js in.aac out.f32 [--gpu-parse] [--tns-spec] [--pns-spec] [--pulses]
 * --gpu-parse: the bitstream parser runs on the device too (GpuFrontEnd), else in JavaScript (FrontEnd).
 * Output: interleaved float32 PCM in [-1, 1).  Needs the built engine (aac.js_amd/csrc/libaacgpu.so + the N-API addon). */
'use strict';
const fs = require('fs'), path = require('path');
const host = require(path.join(__dirname, '..', 'aac.js_amd', 'js'));
const args = process.argv.slice(2), flags = args.filter(function (a) { return a.startsWith('--'); }), files = args.filter(function (a) { return !a.startsWith('--'); });
if (files.length !== 2) { console.error('usage: aacdec.js in.aac out.f32 [--gpu-parse] [--tns-spec] [--pns-spec] [--pulses]'); process.exit(2); }
const gpuParse = flags.includes('--gpu-parse'), tnsSpec = flags.includes('--tns-spec');
const frontend = gpuParse ? new host.GpuFrontEnd({ wantTns: tnsSpec, applyPulses: flags.includes('--pulses'), batch: 4096 }) : new host.FrontEnd();
const dec = new host.GpuAACDecoder({ frontend: frontend, lookahead: 4096,
                                     tnsMode: flags.includes('--tns-spec') ? host.TNS_SPEC : host.TNS_REFERENCE,
                                     pnsMode: flags.includes('--pns-spec') ? host.PNS_SPEC : host.PNS_REFERENCE,
                                     applyPulses: flags.includes('--pulses'), carryWindowShape: flags.includes('--carry-shape') });
dec.init();
const demux = new host.adts.AdtsDemuxer(function (event, payload) {
    if (event === 'format') Object.assign(dec.format, payload);
    else if (event === 'cookie') dec.setCookie(payload);
    else if (event === 'data') dec.feed(payload);
});
demux.push(new Uint8Array(fs.readFileSync(files[0])));
const out = fs.openSync(files[1], 'w');
let frames = 0;
const t0 = Date.now();
for (let pcm; (pcm = dec.readChunk()); frames++) fs.writeSync(out, Buffer.from(pcm.buffer, pcm.byteOffset, pcm.byteLength));
fs.closeSync(out);
console.error(frames + ' frames, ' + dec.format.channelsPerFrame + ' ch, ' + dec.format.sampleRate + ' Hz in ' + (Date.now() - t0) + ' ms');
