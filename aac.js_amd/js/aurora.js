/*
 * aac.js_amd/js/aurora.js — registration with Aurora.js (`av`), the step that makes this package a codec plugin the way
 * the reference is one: src/decoder.js:29-31 (AV.Decoder.extend + register 'mp4a' and 'aac '), src/decoder.js:437
 * (module.exports = the decoder class), src/adts_demuxer.js:4-5 (AV.Demuxer.extend + register).
 *
 *   const AV = require('av');
 *   const { Decoder, Demuxer } = require('aac.js_amd/js/aurora.js').register(AV, { lookahead: 64 });
 *
 * After that AV.Asset / AV.Player pick these classes up for ADTS streams and for 'mp4a' tracks of MP4 files exactly as
 * they would pick up aac.js.  `av` is a peer dependency that is handed in (nothing is required here), as in the
 * reference's package.json:6-8.
 *
 * Decoder: Aurora constructs it with (demuxer, format), appends the demuxer's 'data' buffers to this.stream, and calls
 * init(), setCookie(buffer), readChunk().  The wrapper moves whatever bytes have arrived from this.stream to the
 * bitstream front end (ADTS bytes for formatID 'aac ', the demuxer's buffers of one or more raw_data_blocks for 'mp4a') and serves frames from
 * GpuAACDecoder.readChunk(): a Float32Array of 1024 * channels samples, null when no complete frame is buffered (Aurora
 * then waits for more data and calls again), or the reference's Error for a malformed frame.  Aurora rewinds the
 * stream after a null — bytes already handed to the front end are skipped, not fed twice.  When the demuxer has ended and the
 * last frame has been delivered the decoder releases its slot of a shared engine (options.shared); destroy() does the same early.
 *
 * Demuxer: probe(stream) and the format / cookie / data events of adts_demuxer.js:7-20, 54-80.
 */
'use strict';
const host = require('./index.js');
const adts = require('./adts.js');

function register(AV, options) {
    options = options || {};

    const Decoder = AV.Decoder.extend(function () {
        AV.Decoder.register('mp4a', this);
        AV.Decoder.register('aac ', this);

        this.prototype.init = function () {
            const frontend = options.frontend ? options.frontend() : (options.gpuParse ? new host.GpuFrontEnd(options) : new host.FrontEnd(options));
            /* options.shared: a SharedEngine — every decoder Aurora constructs takes a stream slot of it (cross-stream batches) */
            this.impl = new host.GpuAACDecoder(Object.assign({}, options, { format: this.format, frontend: frontend, engine: options.engine ? options.engine() : null,
                                                                             shared: options.shared || null }));
            this.impl.init();                                   // format.floatingPoint = true (decoder.js:49-51)
            this.fed = 0;                                       // absolute stream offset up to which bytes went to the front end
            this.packets = this.format.formatID === 'mp4a';     // MP4 samples arrive as buffers of whole samples; ADTS is a byte stream
        };

        this.prototype.setCookie = function (buffer) {
            this.impl.setCookie(buffer);
            this.config = this.impl.config;
        };

        this.prototype.readChunk = function () {
            const s = this.stream;
            if (s.offset < this.fed) s.advance(this.fed - s.offset);            // Aurora rewound after a null: those bytes are in already
            while (s.available(1)) {
                const buf = this.packets ? s.readSingleBuffer(s.remainingBytes()) : s.readBuffer(s.remainingBytes());
                if (this.packets) this.impl.feedPacket(buf, true); else this.impl.feed(buf);   // an M4A demuxer's buffer: one sample or a chunk of them
            }
            this.fed = s.offset;
            const out = this.impl.readChunk();
            /* the last buffer is in and nothing is left to deliver: Aurora emits 'end' next and never calls again — the decoder
             * gives its stream slot of a SharedEngine back (otherwise the N-th player of a process would find no slot left) */
            if (out === null && this.receivedFinalBuffer) this.impl.close();
            return out;
        };

        /* hosts that tear a player down early (AV.Asset#stop / destroy) may call this; idempotent */
        this.prototype.destroy = function () { if (this.impl) this.impl.close(); };
    });

    const Demuxer = AV.Demuxer.extend(function () {
        AV.Demuxer.register(this);

        this.probe = function (stream) {
            const offset = stream.offset;
            let found = false;
            while (!found && stream.available(2)) found = (stream.readUInt16() & 0xfff6) === 0xfff0;   // adts_demuxer.js:11-16
            stream.seek(offset);
            return found;
        };

        this.readHeader = adts.readHeader;                      // static, adts_demuxer.js:28-52 (the reference's decoder calls it)

        this.prototype.readChunk = function () {
            if (!this.sentHeader) {
                if (!this.stream.available(7)) return;
                const h = adts.readHeader(new host.BitReader(this.stream.peekBuffer(0, Math.min(9, this.stream.remainingBytes())).data));
                this.emit('format', { formatID: 'aac ', sampleRate: host.SAMPLE_RATES[h.samplingIndex], channelsPerFrame: h.chanConfig, bitsPerChannel: 16 });
                this.emit('cookie', new AV.Buffer(adts.cookie(h)));
                this.sentHeader = true;
            }
            while (this.stream.available(1)) this.emit('data', this.stream.readSingleBuffer(this.stream.remainingBytes()));
        };
    });

    return { Decoder: Decoder, Demuxer: Demuxer };
}

module.exports = { register: register };
