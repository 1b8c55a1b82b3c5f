/*
 * tests/js/probe_provider.js — TEST KIT / build-container tooling, not product code.
 *
 * Recovers a Huffman module's code words by using it as a black box: every codebook's prefix tree is walked by
 * feeding the module's two entry points (decodeScaleFactor(stream), decodeSpectralData(stream, cb, data, off) —
 * the shape of the reference's src/huffman.js:1456-1490) chosen bit strings and watching how many bits each call
 * consumes.  Used by tools/gen/gen_codebooks.js (which writes aac.js_amd/data/aac_codebooks.json) and by
 * tests/js/test_frontend.js to check that the committed tables are what the reference's decoder implements.
 */
'use strict';

const MAX_CODE_LEN = 24;
const UNSIGNED_BOOK = [false, false, false, true, true, false, false, true, true, true, true, true];   // by codebook number

function ProbeStream(prefix) { this.prefix = prefix; this.used = 0; }
ProbeStream.prototype.read = function (n) {
    let v = 0;
    for (let i = 0; i < n; i++, this.used++)
        v = v * 2 + (this.used < this.prefix.length ? this.prefix[this.used] : 0);      // zeros after the prefix
    return v;
};

/* book 0 = scalefactor book.  Returns [[len, code, v...]] sorted by (len, code). */
function probeBook(provider, book) {
    const dim = book === 0 ? 1 : (book < 5 ? 4 : 2), out = [], buf = new Int32Array(4);
    const stack = [[]];
    while (stack.length) {
        const prefix = stack.pop(), s = new ProbeStream(prefix);
        let vals, dead = false;
        try {
            if (book === 0) vals = [provider.decodeScaleFactor(s)];
            else { provider.decodeSpectralData(s, book, buf, 0); vals = Array.prototype.slice.call(buf, 0, dim); }
        } catch (e) { dead = true; }                         // ran off the provider's table: not a code word on this path
        if (dead) { if (prefix.length >= MAX_CODE_LEN) continue; }
        let extra = 0;
        if (!dead && book && UNSIGNED_BOOK[book])
            for (const v of vals) { if (v) extra++; if (book === 11 && Math.abs(v) === 16) extra += 5; }   // sign bit; escape '0' + 4 bits
        const len = dead ? Infinity : s.used - extra;
        if (len <= prefix.length) {
            if (len !== prefix.length) throw new Error('codebook ' + book + ': provider is not a prefix code');
            let code = 0;
            for (const b of prefix) code = code * 2 + b;
            out.push([len, code].concat(vals));             // zero sign bits follow the prefix: unsigned books give magnitudes
        } else {
            if (prefix.length >= MAX_CODE_LEN) throw new Error('codebook ' + book + ': code longer than ' + MAX_CODE_LEN + ' bits');
            stack.push(prefix.concat(1), prefix.concat(0));
        }
    }
    out.sort(function (a, b) { return a[0] - b[0] || a[1] - b[1]; });
    return out;
}

/* { sf: [[len, code, value]...], spectral: [ [[len, code, v0, v1(, v2, v3)]...] x 11 ] } */
function probeAll(provider) {
    if (!provider || typeof provider.decodeScaleFactor !== 'function' || typeof provider.decodeSpectralData !== 'function')
        throw new Error('probe: provider lacks decodeScaleFactor / decodeSpectralData');
    const t = { sf: probeBook(provider, 0), spectral: [] };
    for (let b = 1; b <= 11; b++) t.spectral.push(probeBook(provider, b));
    return t;
}

module.exports = { probeBook, probeAll };
