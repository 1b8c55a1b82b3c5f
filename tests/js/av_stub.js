/*
 * tests/js/av_stub.js — TEST KIT: the part of Aurora.js (`av` ~0.4, the reference's peer dependency, not present in this
 * image) that a codec plugin touches, written from its public behaviour: EventEmitter, Buffer, BufferList, Stream,
 * Demuxer / Decoder base classes with extend / register / find, and the decode loop that calls readChunk(), emits
 * 'data', and rewinds the stream and waits when readChunk() returns nothing.  No arithmetic lives in `av`.
 */
'use strict';

function EventEmitter() { this.events = this.events || {}; }
EventEmitter.prototype.on = function (name, fn) { (this.events[name] = this.events[name] || []).push(fn); return this; };
EventEmitter.prototype.emit = function (name) {
    const args = Array.prototype.slice.call(arguments, 1);
    for (const fn of (this.events[name] || []).slice()) fn.apply(this, args);
};

function Buffer_(data) { this.data = data instanceof Uint8Array ? data : new Uint8Array(data); this.length = this.data.length; }

function BufferList() { this.buffers = []; this.availableBytes = 0; }
BufferList.prototype.append = function (b) { this.buffers.push(b); this.availableBytes += b.length; };

/* a read cursor over the list; `offset` is absolute */
function Stream(list) { this.list = list; this.offset = 0; }
Stream.prototype.total = function () { return this.list.buffers.reduce(function (n, b) { return n + b.length; }, 0); };
Stream.prototype.remainingBytes = function () { return this.total() - this.offset; };
Stream.prototype.available = function (n) { return this.remainingBytes() >= n; };
Stream.prototype.locate = function (pos) {
    for (let i = 0, base = 0; i < this.list.buffers.length; base += this.list.buffers[i].length, i++)
        if (pos < base + this.list.buffers[i].length) return { index: i, at: pos - base };
    return null;
};
Stream.prototype.peekUInt8 = function (off) { const p = this.locate(this.offset + (off | 0)); if (!p) throw new UnderflowError(); return this.list.buffers[p.index].data[p.at]; };
Stream.prototype.readUInt8 = function () { const v = this.peekUInt8(0); this.offset++; return v; };
Stream.prototype.readUInt16 = function () { const hi = this.readUInt8(); return (hi << 8) | this.readUInt8(); };
Stream.prototype.advance = function (n) { if (!this.available(n)) throw new UnderflowError(); this.offset += n; return this; };
Stream.prototype.seek = function (pos) { this.offset = pos; return this; };
Stream.prototype.peekBuffer = function (off, n) {
    const out = new Uint8Array(n);
    for (let i = 0; i < n; i++) out[i] = this.peekUInt8(off + i);
    return new Buffer_(out);
};
Stream.prototype.readBuffer = function (n) { const b = this.peekBuffer(0, n); this.offset += n; return b; };
/* at most n bytes, never across the boundary of the buffers the source delivered */
Stream.prototype.readSingleBuffer = function (n) {
    const p = this.locate(this.offset);
    if (!p) throw new UnderflowError();
    const b = this.list.buffers[p.index], take = Math.min(n, b.length - p.at);
    this.offset += take;
    return new Buffer_(b.data.subarray(p.at, p.at + take));
};

function UnderflowError() { this.name = 'UnderflowError'; }
UnderflowError.prototype = Object.create(Error.prototype);

function makeBase(ctor, keyed) {
    ctor.registry = keyed ? {} : [];
    ctor.extend = function (body) {
        const Base = this;
        function Klass() { Base.apply(this, arguments); }
        Klass.prototype = Object.create(Base.prototype);
        Klass.prototype.constructor = Klass;
        Klass.extend = Base.extend;
        body.call(Klass, Klass);
        return Klass;
    };
    return ctor;
}

/* Demuxer(source, firstChunk): source emits 'data' (Buffer) and 'end' */
const Demuxer = makeBase(function Demuxer(source, chunk) {
    EventEmitter.call(this);
    const list = new BufferList();
    this.stream = new Stream(list);
    let received = false;
    const take = (b) => { list.append(b); received = true; this.readChunk(); };
    source.on('data', take);
    source.on('end', () => { this.emit('end'); });
    if (this.init) this.init();
    if (chunk) take(chunk);
}, false);
Demuxer.prototype = Object.create(EventEmitter.prototype);
Demuxer.register = function (cls) { Demuxer.registry.push(cls); };
Demuxer.find = function (buffer) {
    const list = new BufferList();
    list.append(buffer);
    for (const cls of Demuxer.registry) if (cls.probe(new Stream(list))) return cls;
    return null;
};

/* Decoder(demuxer, format): Aurora's decode loop */
const Decoder = makeBase(function Decoder(demuxer, format) {
    EventEmitter.call(this);
    this.demuxer = demuxer;
    this.format = format;
    const list = new BufferList();
    this.stream = new Stream(list);
    this.receivedFinalBuffer = false;
    this.waiting = false;
    demuxer.on('cookie', (cookie) => { try { this.setCookie(cookie); } catch (e) { this.emit('error', e); } });
    demuxer.on('data', (chunk) => { list.append(chunk); if (this.waiting) this.decode(); });
    demuxer.on('end', () => { this.receivedFinalBuffer = true; if (this.waiting) this.decode(); });
    this.init();
}, true);
Decoder.prototype = Object.create(EventEmitter.prototype);
Decoder.prototype.init = function () {};
Decoder.prototype.setCookie = function () {};
Decoder.prototype.decode = function () {
    this.waiting = false;
    const offset = this.stream.offset;
    let packet = null;
    try { packet = this.readChunk(); }
    catch (error) { if (!(error instanceof UnderflowError)) { this.emit('error', error); return false; } }
    if (packet) { this.emit('data', packet); return true; }
    if (!this.receivedFinalBuffer) { this.stream.seek(offset); this.waiting = true; }
    else this.emit('end');
    return false;
};
Decoder.register = function (id, cls) { Decoder.registry[id] = cls; };
Decoder.find = function (id) { return Decoder.registry[id] || null; };

module.exports = { EventEmitter: EventEmitter, Buffer: Buffer_, BufferList: BufferList, Stream: Stream, UnderflowError: UnderflowError,
                   Demuxer: Demuxer, Decoder: Decoder };
