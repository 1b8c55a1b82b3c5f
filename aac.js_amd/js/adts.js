/*
 * aac.js_amd/js/adts.js — ADTS framing for the JavaScript host: the demuxer side of the plugin surface
 * (reference: src/adts_demuxer.js), written from the ADTS header layout (ISO/IEC 13818-7 6.2 / 14496-3 1.A.2).
 *
 *   probe(bytes)            true if an ADTS syncword is found (the reference's test: 16 bits & 0xfff6 == 0xfff0,
 *                           adts_demuxer.js:7-20)
 *   readHeader(bitReader)   { profile, samplingIndex, chanConfig, frameLength, numFrames, headerBytes } or throws
 *                           'Invalid ADTS header.' (adts_demuxer.js:28-52); consumes the header incl. the CRC
 *   cookie(header)          the 2-byte AudioSpecificConfig the reference synthesises (adts_demuxer.js:66-70)
 *   frames(bytes)           own addition: [{offset, length, header}] of every COMPLETE frame in a buffer, using
 *                           frame_length — what GpuAACDecoder's look-ahead needs to know how many frames it can batch
 *   AdtsDemuxer             format / cookie once, then data, through a callback (adts_demuxer.js:54-78)
 *
 * Own code; nothing is imported from the reference.
 */
'use strict';
const { BitReader, SAMPLE_RATES } = require('./index.js');

function probe(bytes) {
    for (let i = 0; i + 1 < bytes.length; i += 2)       // the reference steps by readUInt16: even offsets only
        if ((((bytes[i] << 8) | bytes[i + 1]) & 0xfff6) === 0xfff0) return true;
    return false;
}

function readHeader(r) {
    if (r.read(12) !== 0xfff) throw new Error('Invalid ADTS header.');
    r.advance(3);                                        // MPEG version, layer
    const protectionAbsent = r.read(1) === 1;
    const h = {};
    h.profile = r.read(2) + 1;
    h.samplingIndex = r.read(4);
    r.advance(1);                                        // private bit
    h.chanConfig = r.read(3);
    r.advance(4);                                        // original/copy, home, copyright id bit + start
    h.frameLength = r.read(13);                          // header included
    r.advance(11);                                       // buffer fullness
    h.numFrames = r.read(2) + 1;
    if (!protectionAbsent) r.advance(16);                // CRC
    h.headerBytes = protectionAbsent ? 7 : 9;
    return h;
}

function cookie(h) {
    return new Uint8Array([(h.profile << 3) | ((h.samplingIndex >> 1) & 7), ((h.samplingIndex & 1) << 7) | (h.chanConfig << 3)]);
}

function frames(bytes) {
    const out = [];
    let off = 0;
    while (off + 7 <= bytes.length) {
        const r = new BitReader(bytes.subarray(off));
        let h;
        try { h = readHeader(r); } catch (e) { break; }
        if (h.frameLength < h.headerBytes || off + h.frameLength > bytes.length) break;   // incomplete: wait for more data
        out.push({ offset: off, length: h.frameLength, header: h });
        off += h.frameLength;
    }
    return out;
}

/* emit(event, payload): 'format' {formatID, sampleRate, channelsPerFrame, bitsPerChannel}, 'cookie' Uint8Array(2),
 * 'data' Uint8Array — the events the reference's demuxer raises, in the same order */
function AdtsDemuxer(emit) { this.emit = emit; this.sentHeader = false; }
AdtsDemuxer.probe = probe;
AdtsDemuxer.readHeader = readHeader;
AdtsDemuxer.prototype.push = function (bytes) {
    if (!this.sentHeader) {
        const h = readHeader(new BitReader(bytes));
        this.emit('format', { formatID: 'aac ', sampleRate: SAMPLE_RATES[h.samplingIndex], channelsPerFrame: h.chanConfig, bitsPerChannel: 16 });
        this.emit('cookie', cookie(h));
        this.sentHeader = true;
    }
    if (bytes.length) this.emit('data', bytes);
};

module.exports = { probe, readHeader, cookie, frames, AdtsDemuxer };
