/*
 * aac.js_amd/js/codebooks.js — the 12 AAC Huffman codebooks as two-level lookup tables.
 *
 * The code words (ISO/IEC 14496-3 tables 4.A.1-4.A.12; the reference keeps the same facts as the private arrays
 * HCB1..HCB11 / HCB_SF of src/huffman.js:22-1418) ship with this package in its own format:
 * aac.js_amd/data/aac_codebooks.json — per book, (length, code word) in the standard's index order, the values
 * following from the index (tools/gen/gen_codebooks.js documents the layout and wrote the file).
 *
 *   standard()        the shipped books (built once, cached)
 *   fromTables(t)     t = { sf: [[len, code, value]...], spectral: [ [[len, code, v0, v1(, v2, v3)]...] x 11 ] }:
 *                     any complete prefix codes over the same alphabets (the tests use stand-ins to exercise the
 *                     table builder; toTables() writes this shape)
 *   load(opts)        opts.codebooks / opts.tables, else standard()
 *
 * What is built from the entries is this repository's own decoder: a 9-bit first-level table and second-level
 * tables for the longer codes (one peek + one table read per code word; the reference compares code words one
 * by one down a list sorted by length, huffman.js:1426-1439).
 *
 * Facts of the standard used here: books 1,2,5,6 carry signed values, the others magnitudes followed by one sign
 * bit per non-zero value; books 1-4 code 4 values, 5-11 code 2; book 11 has the escape (|v| = 16 -> N ones, a
 * zero, N+4 bits; value 2^(N+4) + bits).
 */
'use strict';
const fs = require('fs'), path = require('path');

const L1_BITS = 9;
const UNSIGNED_BOOK = [false, false, false, true, true, false, false, true, true, true, true, true];   // by codebook number
const ENTRY_COUNT = [121, 81, 81, 81, 81, 81, 81, 64, 64, 169, 169, 289];                              // sf, 1..11
const DATA_FILE = path.join(__dirname, '..', 'data', 'aac_codebooks.json');

/* aac_codebooks.json -> the { sf, spectral } shape, each book sorted by (length, code word) */
function tablesFromData(data) {
    if (!data || !Array.isArray(data.books) || data.books.length !== 12) throw new Error('codebooks: malformed data file');
    const out = data.books.map(function (b, book) {
        const n = Math.pow(b.mod, b.dim), off = b.signed ? (b.mod - 1) / 2 : 0, entries = [];
        if (b.book !== book || n !== ENTRY_COUNT[book] || b.codes.length !== 6 * n) throw new Error('codebooks: malformed book ' + book);
        for (let idx = 0; idx < n; idx++) {
            const w = parseInt(b.codes.substr(6 * idx, 6), 16), e = [w >>> 19, w & 0x7ffff];
            for (let j = b.dim - 1, r = idx; j >= 0; j--, r = Math.floor(r / b.mod)) e[2 + j] = r % b.mod - off;      // first value most significant
            entries.push(e);
        }
        return entries.sort(function (a, c) { return a[0] - c[0] || a[1] - c[1]; });
    });
    return { sf: out[0], spectral: out.slice(1) };
}

/* ---- two-level tables --------------------------------------------------------------------------------------
 * entry >= 0: (symbol << 5) | length;   entry < 0: ~entry = (offset << 5) | extra bits of a second-level table */
function buildLookup(entries, book) {
    let kraft = 0;
    for (const e of entries) kraft += Math.pow(2, -e[0]);
    if (entries.length !== ENTRY_COUNT[book] || Math.abs(kraft - 1) > 1e-12)
        throw new Error('codebook ' + book + ': ' + entries.length + ' entries, Kraft sum ' + kraft + ' (expected ' + ENTRY_COUNT[book] + ', 1)');
    const l1 = new Int32Array(1 << L1_BITS), deep = new Map();
    entries.forEach(function (e, sym) {
        const len = e[0], code = e[1];
        if (len <= L1_BITS) {
            const lo = code << (L1_BITS - len);
            for (let i = 0; i < (1 << (L1_BITS - len)); i++) l1[lo + i] = (sym << 5) | len;
        } else {
            const head = Math.floor(code / Math.pow(2, len - L1_BITS));
            if (!deep.has(head)) deep.set(head, []);
            deep.get(head).push(sym);
        }
    });
    let size = 0;
    deep.forEach(function (syms) {
        let extra = 0;
        for (const s of syms) extra = Math.max(extra, entries[s][0] - L1_BITS);
        size += 1 << extra;
    });
    const l2 = new Int32Array(size);
    let off = 0;
    deep.forEach(function (syms, head) {
        let extra = 0;
        for (const s of syms) extra = Math.max(extra, entries[s][0] - L1_BITS);
        for (const s of syms) {
            const len = entries[s][0], tail = entries[s][1] % Math.pow(2, len - L1_BITS), lo = tail << (extra - (len - L1_BITS));
            for (let i = 0; i < (1 << (extra - (len - L1_BITS))); i++) l2[off + lo + i] = (s << 5) | len;
        }
        l1[head] = ~((off << 5) | extra);
        off += 1 << extra;
    });
    return { l1: l1, l2: l2 };
}

function Codebooks(tables) {
    this.tables = tables;
    this.look = [buildLookup(tables.sf, 0)];
    this.values = [Int16Array.from(tables.sf, function (e) { return e[2]; })];
    for (let b = 1; b <= 11; b++) {
        const ent = tables.spectral[b - 1], dim = b < 5 ? 4 : 2;
        this.look.push(buildLookup(ent, b));
        const v = new Int8Array(ent.length * dim);
        ent.forEach(function (e, s) { for (let j = 0; j < dim; j++) v[s * dim + j] = e[2 + j]; });
        this.values.push(v);
    }
}

/* index of the next code word of `book` in `bits` (a BitStream), consuming it */
Codebooks.prototype.symbol = function (bits, book) {
    const t = this.look[book];
    let e = t.l1[bits.peek25(L1_BITS)];
    if (e < 0) {
        const ptr = ~e, extra = ptr & 31;
        e = t.l2[(ptr >>> 5) + (bits.peek25(L1_BITS + extra) & ((1 << extra) - 1))];
    }
    bits.advance(e & 31);
    return e >>> 5;
};

/* the scalefactor book's value, 0..120 (huffman.js:1456-1459) */
Codebooks.prototype.scaleFactor = function (bits) { return this.values[0][this.symbol(bits, 0)]; };

/* one code word of spectral book 1..11 -> 4 or 2 quantised values at q[off..] (huffman.js:1462-1490) */
Codebooks.prototype.spectral = function (bits, book, q, off) {
    const dim = book < 5 ? 4 : 2, v = this.values[book], s = this.symbol(bits, book) * dim;
    if (!UNSIGNED_BOOK[book]) {
        for (let j = 0; j < dim; j++) q[off + j] = v[s + j];
        return;
    }
    for (let j = 0; j < dim; j++) {
        const m = v[s + j];
        q[off + j] = m && bits.read1() ? -m : m;
    }
    if (book === 11)
        for (let j = 0; j < 2; j++) {
            const m = q[off + j];
            if (m !== 16 && m !== -16) continue;
            let n = 4;
            while (bits.read1()) n++;
            if (n > 12) throw new Error('Huffman: escape sequence too long');      // the standard stops at 8 ones: 13 bits, |v| <= 8191
            const mag = (1 << n) + bits.read(n);
            q[off + j] = m < 0 ? -mag : mag;
        }
};

Codebooks.prototype.toTables = function () { return this.tables; };

/* the books as aacg_code_entry records (include/aacgpu.h: u32 code, u8 len, i8 v[4], 3 pad) for aacg_parser_create */
Codebooks.prototype.toEntryRecords = function () {
    const books = [this.tables.sf].concat(this.tables.spectral), counts = new Uint32Array(12);
    let n = 0;
    books.forEach(function (b, i) { counts[i] = b.length; n += b.length; });
    const entries = new Uint8Array(12 * n), view = new DataView(entries.buffer);
    let o = 0;
    for (const b of books)
        for (const e of b) {
            view.setUint32(o, e[1], true); view.setUint8(o + 4, e[0]);
            for (let j = 0; j < e.length - 2; j++) view.setInt8(o + 5 + j, e[2 + j]);
            o += 12;
        }
    return { entries: entries, counts: counts };
};

function fromTables(t) {
    if (!t || !Array.isArray(t.sf) || !Array.isArray(t.spectral) || t.spectral.length !== 11) throw new Error('codebooks: malformed table object');
    return new Codebooks(t);
}

let cached = null;
function standard() {
    if (!cached) cached = new Codebooks(tablesFromData(JSON.parse(fs.readFileSync(DATA_FILE, 'utf8'))));
    return cached;
}

function load(opts) {
    opts = opts || {};
    if (opts.codebooks) return opts.codebooks;
    if (opts.tables) return fromTables(opts.tables);
    return standard();
}

module.exports = { Codebooks, standard, fromTables, tablesFromData, load, UNSIGNED_BOOK, ENTRY_COUNT };
