/*
 * aac.js_amd/js/bits.js — MSB-first bit reader for the frame parser (the role AV.Bitstream plays for the
 * reference: read / peek / advance / align, decoder.js:126-201), over one Uint8Array.
 *
 * read(n) and peek(n) take n <= 32.  Reading past the end throws 'Insufficient data' (what AV.Bitstream does
 * on underflow); peeking past the end sees zero bits, so that a table lookup may look ahead of the last code
 * word — the following advance() is what detects the overrun.
 *
 * Own code; nothing is imported from the reference.
 */
'use strict';

function BitStream(bytes, bitOffset) {
    this.bytes = bytes;
    this.pos = bitOffset | 0;             // in bits
    this.end = bytes.length * 8;
}

/* up to 25 bits at the current position, no bounds check */
BitStream.prototype.peek25 = function (n) {
    const b = this.bytes, p = this.pos, i = p >>> 3;
    const w = ((b[i] << 24) | (b[i + 1] << 16) | (b[i + 2] << 8) | b[i + 3]) << (p & 7);   // bytes past the end: undefined -> 0
    return w >>> (32 - n);
};

BitStream.prototype.peek = function (n) {
    if (n === 0) return 0;
    if (n <= 25) return this.peek25(n);
    const hi = this.peek25(n - 16);
    this.pos += n - 16;
    const lo = this.peek25(16);
    this.pos -= n - 16;
    return (hi * 65536 + lo) >>> 0;
};

BitStream.prototype.advance = function (n) {
    this.pos += n;
    if (this.pos > this.end) throw new Error('Insufficient data');
};

BitStream.prototype.read = function (n) {
    const v = this.peek(n);
    this.advance(n);
    return v;
};

BitStream.prototype.read1 = function () {
    const p = this.pos;
    if (p >= this.end) throw new Error('Insufficient data');
    this.pos = p + 1;
    return (this.bytes[p >>> 3] >>> (7 - (p & 7))) & 1;
};

BitStream.prototype.align = function () { this.pos = (this.pos + 7) & ~7; };
BitStream.prototype.available = function (n) { return this.pos + n <= this.end; };
BitStream.prototype.bitsLeft = function () { return this.end - this.pos; };

module.exports = { BitStream };
