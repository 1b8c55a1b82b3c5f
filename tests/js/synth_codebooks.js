/*
 * tests/js/synth_codebooks.js — TEST KIT: stand-in Huffman codebooks for machines without the real code words
 * (they are not part of this repository, see aac.js_amd/js/codebooks.js).
 *
 * Same alphabets as the standard's 12 books (ISO/IEC 14496-3 4.A.1: 121 scalefactor steps; {-1..1}^4, {0..2}^4,
 * {-4..4}^2, {0..7}^2, {0..12}^2, {0..16}^2), code words of my own making: a Huffman tree over seeded random
 * weights that favour small values, depth-limited to 19 bits, canonical assignment.  A parser does not care
 * which prefix code it is given, so writer -> parser tests run on these where the real tables are absent; the
 * stream bytes they produce are of course not AAC.
 */
'use strict';
const { Rng } = require('./aac_writer.js');

function alphabet(book) {
    if (book === 0) { const a = []; for (let i = 0; i <= 120; i++) a.push([i]); return a; }
    const dim = book < 5 ? 4 : 2;
    const lo = book <= 2 ? -1 : book === 5 || book === 6 ? -4 : 0, hi = [0, 1, 1, 2, 2, 4, 4, 7, 7, 12, 12, 16][book];
    let out = [[]];
    for (let d = 0; d < dim; d++) { const next = []; for (const p of out) for (let v = lo; v <= hi; v++) next.push(p.concat(v)); out = next; }
    return out;
}

/* code lengths of a Huffman tree over `weights` */
function huffmanLengths(weights) {
    let nodes = weights.map(function (w, i) { return { w: w, leaves: [i] }; });
    const len = new Array(weights.length).fill(0);
    while (nodes.length > 1) {
        nodes.sort(function (a, b) { return a.w - b.w; });
        const a = nodes.shift(), b = nodes.shift();
        for (const i of a.leaves) len[i]++;
        for (const i of b.leaves) len[i]++;
        nodes.push({ w: a.w + b.w, leaves: a.leaves.concat(b.leaves) });
    }
    return len;
}

function book(rng, b) {
    const values = alphabet(b);
    for (let flat = 1.0; ; flat *= 0.8) {                       // flatter weights until the tree is shallow enough
        const weights = values.map(function (v) {
            const size = b === 0 ? Math.abs(v[0] - 60) : v.reduce(function (s, x) { return s + Math.abs(x); }, 0);
            return Math.exp(-flat * (0.35 * size + 2 * (rng.next() / 4294967296)));
        });
        const len = huffmanLengths(weights);
        if (Math.max.apply(null, len) > 19) continue;
        const order = values.map(function (_, i) { return i; }).sort(function (x, y) { return len[x] - len[y] || x - y; });
        const entries = [];
        let code = 0, prev = len[order[0]];
        for (const i of order) {
            code = code * Math.pow(2, len[i] - prev); prev = len[i];
            entries.push([len[i], code].concat(values[i]));
            code++;
        }
        return entries;
    }
}

function synthTables(seed) {
    const rng = new Rng(seed || 0xC0DE);
    const t = { sf: book(rng, 0), spectral: [] };
    for (let b = 1; b <= 11; b++) t.spectral.push(book(rng, b));
    return t;
}

module.exports = { synthTables, alphabet };
