#!/usr/bin/env node
/*
 * aac_port.js — JavaScript restatement of the transform path, for the CPU baseline only.
 *
 * TEST / BASELINE INFRASTRUCTURE (lives under oracle/): never imported by the product.  It stands in for
 * "aac.js's own Node path" on the GPU box, where the reference itself cannot travel: same algorithm class
 * as the reference (table dequant, MS, pre-twiddle + N/4 complex FFT + post-twiddle IMDCT, window,
 * overlap-add, interleave; src/ics.js:244-256, src/decoder.js:379-404,203-215, src/mdct.js:62-115,
 * src/fft.js:105-192, src/filter_bank.js:105-118), written from scratch on flat typed arrays.
 * ONLY_LONG_SEQUENCE stereo frames (BASELINE config 2).  Checked against tests/golden in
 * tests/test_js_host.py::test_js_port_matches_golden; timed by bench.py (cpu_baseline.js_port).
 *
 *   node oracle/js/aac_port.js bench <seconds>     -> JSON {frames_per_s, frames, seconds}
 *   node oracle/js/aac_port.js check <golden dir>  -> JSON {rms}
 */
'use strict';

const N = 2048, N2 = 1024, N4 = 512, N8 = 256;

// ---- tables (by formula; tables.js:168-191, filter_bank.js:46-79, mdct_tables.js) ----
const IQ = new Float32Array(8191), SF = new Float32Array(428);
for (let i = 0; i < 8191; i++) IQ[i] = Math.pow(i, 4 / 3);
for (let i = 0; i < 428; i++) SF[i] = Math.pow(2, (i - 200) / 4);
const SINE = new Float32Array(1024), KBD = new Float32Array(1024);
for (let i = 0; i < 1024; i++) SINE[i] = Math.sin((i + 0.5) * (Math.PI / 2048));
(function () {
    const a2 = (4 * Math.PI / 1024) * (4 * Math.PI / 1024), f = new Float32Array(1024);
    let sum = 0;
    for (let n = 0; n < 1024; n++) {
        const t = n * (1024 - n) * a2;
        let b = 1;
        for (let j = 50; j > 0; j--) b = b * t / (j * j) + 1;
        sum += b; f[n] = sum;
    }
    sum++;
    for (let n = 0; n < 1024; n++) KBD[n] = Math.sqrt(f[n] / sum);
})();
const COS = new Float64Array(N4), SIN = new Float64Array(N4);
for (let k = 0; k < N4; k++) {
    const a = 2 * Math.PI * (k + 0.125) / N, s = Math.sqrt(2 / N);
    COS[k] = s * Math.cos(a); SIN[k] = s * Math.sin(a);
}
const WR = new Float64Array(N4 / 2), WI = new Float64Array(N4 / 2);
for (let k = 0; k < N4 / 2; k++) { WR[k] = Math.cos(2 * Math.PI * k / N4); WI[k] = Math.sin(2 * Math.PI * k / N4); }
const REV = new Uint16Array(N4);
for (let i = 0; i < N4; i++) { let r = 0; for (let b = 0; b < 9; b++) if (i & (1 << b)) r |= 1 << (8 - b); REV[i] = r; }
const SWB48 = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 56, 64, 72, 80, 88, 96, 108, 120, 132, 144, 160, 176, 196, 216, 240, 264,
               292, 320, 352, 384, 416, 448, 480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800, 832, 864, 896, 928, 1024];

// ---- the path for one ONLY_LONG stereo frame -------------------------------------------
const re = new Float32Array(N4), im = new Float32Array(N4), buf = new Float32Array(N);

function dequant(q, qOff, meta, mOff, maxSFB, out) {
    out.fill(0);
    for (let sfb = 0; sfb < maxSFB; sfb++) {
        const w = meta[mOff + sfb], bt = w >> 12;
        if (bt === 0 || bt >= 13) continue;
        const sf = (w & 0x200) ? -SF[w & 0x1ff] : SF[w & 0x1ff];
        for (let k = SWB48[sfb]; k < SWB48[sfb + 1]; k++) {
            const v = q[qOff + k];
            out[k] = (v > 0 ? IQ[v] : -IQ[-v]) * sf;
        }
    }
}

function midSide(metaL, mL, metaR, mR, maxSFB, l, r) {
    for (let sfb = 0; sfb < maxSFB; sfb++) {
        if (!(metaL[mL + sfb] & 0x400) || (metaL[mL + sfb] >> 12) >= 13 || (metaR[mR + sfb] >> 12) >= 13) continue;
        for (let k = SWB48[sfb]; k < SWB48[sfb + 1]; k++) { const t = l[k] - r[k]; l[k] = l[k] + r[k]; r[k] = t; }
    }
}

function imdctLong(x, out) {
    for (let k = 0; k < N4; k++) {
        const a = x[2 * k], b = x[N2 - 1 - 2 * k], j = REV[k];
        im[j] = a * COS[k] + b * SIN[k];
        re[j] = b * COS[k] - a * SIN[k];
    }
    for (let half = 1; half < N4; half <<= 1) {           // radix-2 decimation in time, e^{+i theta}
        const step = N4 / (2 * half);
        for (let j = 0; j < N4; j += 2 * half)
            for (let k = 0; k < half; k++) {
                const wr = WR[k * step], wi = WI[k * step], p = j + k, s = p + half;
                const zr = re[s] * wr - im[s] * wi, zi = re[s] * wi + im[s] * wr;
                re[s] = re[p] - zr; im[s] = im[p] - zi;
                re[p] += zr; im[p] += zi;
            }
    }
    for (let k = 0; k < N4; k++) {
        const r0 = re[k], i0 = im[k];
        im[k] = i0 * COS[k] + r0 * SIN[k];
        re[k] = r0 * COS[k] - i0 * SIN[k];
    }
    for (let k = 0; k < N8; k++) {
        out[2 * k] = im[N8 + k];            out[2 * k + 1] = -re[N8 - 1 - k];
        out[N4 + 2 * k] = re[k];            out[N4 + 2 * k + 1] = -im[N4 - 1 - k];
        out[N2 + 2 * k] = re[N8 + k];       out[N2 + 2 * k + 1] = -im[N8 - 1 - k];
        out[N2 + N4 + 2 * k] = -im[k];      out[N2 + N4 + 2 * k + 1] = re[N4 - 1 - k];
    }
}

function filterbankLong(x, shape, shapePrev, overlap, pcm, pOff, C) {
    const w = shape ? KBD : SINE, wp = shapePrev ? KBD : SINE;
    imdctLong(x, buf);
    for (let i = 0; i < N2; i++) pcm[pOff + i * C] = (overlap[i] + buf[i] * wp[i]) / 32768;
    for (let i = 0; i < N2; i++) overlap[i] = buf[N2 + i] * w[N2 - 1 - i];
}

const L = new Float32Array(N2), R = new Float32Array(N2);
function decodeStereoFrame(q, qOff, meta, mOff, shape, ovL, ovR, pcm, pOff, ms) {
    dequant(q, qOff, meta, mOff, 49, L);
    dequant(q, qOff + 1024, meta, mOff + 120, 49, R);
    if (ms) midSide(meta, mOff, meta, mOff + 120, 49, L, R);
    filterbankLong(L, shape, 0, ovL, pcm, pOff, 2);
    filterbankLong(R, shape, 0, ovR, pcm, pOff + 1, 2);
}

module.exports = { decodeStereoFrame };

if (require.main === module) {
    const mode = process.argv[2] || 'bench';
    if (mode === 'bench') {
        const seconds = parseFloat(process.argv[3] || '5');
        // 16 frames of synthetic config-2 data (xorshift32; same shape of magnitudes as the GPU workload)
        let s = 0xAAC00002 >>> 0;
        const rnd = function () { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
        const T = 16, q = new Int16Array(T * 2048), meta = new Uint16Array(T * 240), pcm = new Float32Array(T * 2048);
        for (let i = 0; i < q.length; i++) { const k = i & 1023, amp = Math.max(1, Math.floor(12 * Math.exp(-k / 180))); q[i] = (rnd() % (2 * amp + 1)) - amp; }
        for (let t = 0; t < 2 * T; t++) for (let b = 0; b < 49; b++) meta[t * 120 + b] = (240 + rnd() % 17) | ((1 + rnd() % 11) << 12) | ((b & 1) ? 0 : 0x400);
        const ovL = new Float32Array(1024), ovR = new Float32Array(1024);
        let frames = 0;
        const t0 = process.hrtime.bigint();
        let dt = 0;
        do {
            for (let t = 0; t < T; t++) decodeStereoFrame(q, t * 2048, meta, t * 240, 1, ovL, ovR, pcm, t * 2048, true);
            frames += T;
            dt = Number(process.hrtime.bigint() - t0) / 1e9;
        } while (dt < seconds);
        console.log(JSON.stringify({ frames_per_s: frames / dt, frames: frames, seconds: dt, node: process.version }));
    } else {
        // check against a golden scenario: ONLY_LONG frames of scn_stereo (common window)
        const fs = require('fs'), path = require('path'), dir = process.argv[3];
        const man = JSON.parse(fs.readFileSync(path.join(dir, 'golden.json'))), blob = fs.readFileSync(path.join(dir, 'golden.bin'));
        const arr = function (name, Ty) { const a = man.arrays[name], n = a.shape.reduce((x, y) => x * y, 1); const c = new Uint8Array(n * Ty.BYTES_PER_ELEMENT); c.set(blob.subarray(a.offset, a.offset + c.length)); return new Ty(c.buffer); };
        const spec = arr('scn_stereo.spec', Float32Array), ref = arr('scn_stereo.pcm', Float32Array), units = arr('scn_stereo.units', Uint8Array);
        // frames 0 and 1 are ONLY_LONG: run the filterbank seam on their spectra
        const ov = [new Float32Array(1024), new Float32Array(1024)], out = new Float32Array(2 * 2048);
        let err = 0, cnt = 0;
        for (let t = 0; t < 2; t++) {
            if (units[64 * t + 24] !== 0) throw new Error('fixture frame ' + t + ' is not ONLY_LONG');
            for (let c = 0; c < 2; c++) filterbankLong(spec.subarray((2 * t + c) * 1024, (2 * t + c + 1) * 1024), units[64 * t + 24 + 16 * c + 1], 0, ov[c], out, t * 2048 + c, 2);
            for (let i = 0; i < 2048; i++) { const d = out[t * 2048 + i] - ref[t * 2048 + i]; err += d * d; cnt++; }
        }
        console.log(JSON.stringify({ rms: Math.sqrt(err / cnt) }));
    }
}
