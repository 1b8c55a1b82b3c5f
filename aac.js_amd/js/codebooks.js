/*
 * aac.js_amd/js/codebooks.js — the 12 AAC Huffman codebooks as two-level lookup tables.
 *
 * The code words themselves (ISO/IEC 14496-3 tables 4.A.1-4.A.12; the reference keeps them as the private
 * arrays HCB1..HCB11 / HCB_SF of src/huffman.js:22-1418) are NOT part of this repository.  They come from a
 * *provider* at start-up:
 *
 *   fromProvider(h)   h is any object with the two entry points of the reference's Huffman module
 *                     (decodeScaleFactor(stream), decodeSpectralData(stream, cb, data, off); huffman.js:1456-1490)
 *                     — in a drop-in installation that is require('aac/src/huffman').  The provider is used as
 *                     a black box: every codebook's prefix tree is walked once by feeding it chosen bit strings
 *                     and watching how many bits it consumes, which yields (length, code word, values) for
 *                     every entry.  After that the provider is not called again.
 *   fromTables(t)     t = { sf: [[len, code, value]...], spectral: [ [[len, code, v0, v1(, v2, v3)]...] x 11 ] },
 *                     e.g. a file the user generated from the standard's tables (toTables() writes this shape)
 *   load(opts)        opts.huffman / opts.tables / $AACG_HUFFMAN_MODULE / an installed `aac` package, in that
 *                     order; throws if none is there — the parser cannot run without code words
 *
 * What is built from the entries is this repository's own decoder: a 9-bit first-level table and second-level
 * tables for the longer codes (one peek + one table read per code word; the reference compares code words one
 * by one down a list sorted by length, huffman.js:1426-1439).
 *
 * Facts of the standard used here (not taken from the provider): books 1,2,5,6 carry signed values, the others
 * magnitudes followed by one sign bit per non-zero value; books 1-4 code 4 values, 5-11 code 2; book 11 has the
 * escape (|v| = 16 -> N ones, a zero, N+4 bits; value 2^(N+4) + bits).
 */
'use strict';

const L1_BITS = 9, MAX_CODE_LEN = 24;
const UNSIGNED_BOOK = [false, false, false, true, true, false, false, true, true, true, true, true];   // by codebook number
const ENTRY_COUNT = [121, 81, 81, 81, 81, 81, 81, 64, 64, 169, 169, 289];                              // sf, 1..11

/* ---- walking a provider's prefix trees ------------------------------------------------------------------ */
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
            out.push([len, code].concat(vals));
        } else {
            if (prefix.length >= MAX_CODE_LEN) throw new Error('codebook ' + book + ': code longer than ' + MAX_CODE_LEN + ' bits');
            stack.push(prefix.concat(1), prefix.concat(0));
        }
    }
    out.sort(function (a, b) { return a[0] - b[0] || a[1] - b[1]; });
    return out;
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

function fromProvider(h) {
    if (!h || typeof h.decodeScaleFactor !== 'function' || typeof h.decodeSpectralData !== 'function')
        throw new Error('codebooks: provider lacks decodeScaleFactor / decodeSpectralData');
    const t = { sf: probeBook(h, 0), spectral: [] };
    for (let b = 1; b <= 11; b++) t.spectral.push(probeBook(h, b));
    return new Codebooks(t);
}

function load(opts) {
    opts = opts || {};
    if (opts.codebooks) return opts.codebooks;
    if (opts.tables) return fromTables(opts.tables);
    if (opts.huffman) return fromProvider(opts.huffman);
    const tried = [];
    for (const name of [process.env.AACG_HUFFMAN_MODULE, 'aac/src/huffman', 'aac/src/huffman.js']) {
        if (!name) continue;
        let mod = null;
        try { mod = require(name); } catch (e) { tried.push(name); continue; }
        return fromProvider(mod);
    }
    throw new Error('AAC Huffman codebooks unavailable (tried ' + tried.join(', ') + '): pass { huffman: require("aac/src/huffman") }, ' +
                    '{ tables: ... } or set AACG_HUFFMAN_MODULE; the frame parser cannot run without them');
}

module.exports = { Codebooks, fromProvider, fromTables, load, probeBook, UNSIGNED_BOOK, ENTRY_COUNT };
