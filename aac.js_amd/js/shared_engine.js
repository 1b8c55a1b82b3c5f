/*
 * aac.js_amd/js/shared_engine.js — one engine for many decoders: cross-stream batching behind the plugin surface.
 *
 * The reference is one AACDecoder per stream (src/decoder.js:125-216): N concurrent Aurora players are N decoders, and a
 * GpuAACDecoder on its own hands the GPU its OWN stream's look-ahead (16 frames = one workgroup of a 256-CU chip).  Decoders
 * created with { shared: sharedEngine } instead take a stream slot each of an engine they share — one engine per sample rate,
 * which fixes the scalefactor-band tables — and whenever one of them runs out of decoded frames, readChunk() has EVERY
 * registered decoder parse ahead what it has buffered and submits all of it as ONE batch:
 *
 *   const shared = new SharedEngine({ maxStreams: 256, maxChannels: 8 });
 *   const dec = new GpuAACDecoder({ shared, frontend, lookahead: 16 });  ...  dec.readChunk()
 *
 * The surface stays the reference's (init / setCookie / readChunk, synchronous); a stream's frames still come back in order,
 * a malformed frame's error is still thrown by the readChunk call that reaches it, and a stream's PCM does not depend on
 * which other streams shared its batches (streams only meet in the launch: no arithmetic crosses them).  An engine error
 * on a shared batch is retried decoder by decoder, so that it only costs the stream that caused it.
 */
'use strict';
const host = require('./index.js');

function SharedEngine(opts) {
    opts = opts || {};
    this.opts = opts;
    this.maxStreams = opts.maxStreams || 256;
    this.maxChannels = opts.maxChannels || 8;
    this.groups = new Map();                              // sampleIndex -> { engine, decoders: [], free: [] }
    this.stats = { batches: 0, frames: 0, units: 0, engineNs: 0n, retries: 0 };
}

/* called by GpuAACDecoder.setCookie: the engine for the decoder's sample rate and a stream slot in it */
SharedEngine.prototype.attach = function (dec) {
    const cfg = dec.config;
    if (cfg.chanConfig + dec.maxCoupling > this.maxChannels) throw new Error('SharedEngine: ' + cfg.chanConfig + ' channels exceed maxChannels ' + this.maxChannels);
    if (dec.group) this.detach(dec);                       // a new cookie: a new stream
    let g = this.groups.get(cfg.sampleIndex);
    if (!g) {
        const o = this.opts;
        const engine = o.engine ? o.engine(cfg.sampleIndex) :
            new host.Engine({ sampleIndex: cfg.sampleIndex, maxStreams: this.maxStreams, maxChannels: this.maxChannels, inputKind: host.INPUT_QUANT_I16,
                              deviceOrdinal: o.deviceOrdinal | 0, tnsMode: o.tnsMode | 0, pnsMode: o.pnsMode | 0, cceMode: o.cceMode | 0 });
        g = { engine: engine, decoders: [], free: [], next: 0 };
        this.groups.set(cfg.sampleIndex, g);
    }
    if ((dec.tnsMode | 0) !== (this.opts.tnsMode | 0) || (dec.pnsMode | 0) !== (this.opts.pnsMode | 0) || (dec.cceMode | 0) !== (this.opts.cceMode | 0))
        throw new Error('SharedEngine: the decoder\'s tnsMode / pnsMode / cceMode differ from the shared engine\'s');
    const slot = g.free.length ? g.free.pop() : g.next++;
    if (slot >= this.maxStreams) { g.next--; throw new Error('SharedEngine: more than ' + this.maxStreams + ' streams at one sample rate'); }
    dec.engine = g.engine; dec.stream = slot; dec.group = g;
    g.decoders.push(dec);
};

SharedEngine.prototype.detach = function (dec) {
    const g = dec.group;
    if (!g) return;
    g.decoders.splice(g.decoders.indexOf(dec), 1);
    g.free.push(dec.stream);
    dec.group = null;
};

/* every registered decoder parses ahead what it has buffered; one batch per engine */
SharedEngine.prototype.flush = function () {
    for (const g of this.groups.values()) this.flushGroup(g);
};

SharedEngine.prototype.flushGroup = function (g) {
    const tnsList = (this.opts.tnsMode | 0) === host.TNS_SPEC ? [] : null, cceList = (this.opts.cceMode | 0) === host.CCE_SPEC ? [] : null;
    const parts = [];
    let block = 0, pcmAt = 0;
    for (const dec of g.decoders) {
        let part;
        try { part = dec.collectAhead(block, pcmAt, tnsList, cceList); }
        catch (err) { dec.queue.push(err instanceof Error ? err : new Error(String(err))); continue; }    // unsupported profile: that decoder's own error
        if (!part.frames.length) { dec.deliver(part, null, 0, null); continue; }
        part.dec = dec; part.blockBase = block; part.pcmBase = pcmAt;
        /* where this part's TNS / coupling records sit in the shared lists is already in its units */
        parts.push(part);
        block += part.nBlocks;
        pcmAt += part.frames.length * 1024 * dec.config.chanConfig;
    }
    if (!parts.length) return;
    const q = new Int16Array(block * 1024), meta = new Uint16Array(block * host.META_WORDS), pcm = new Float32Array(pcmAt);
    let units = [];
    for (const p of parts) { p.dec.fillBatch(p, p.blockBase, q, meta); units = units.concat(p.units); }
    const tns = tnsList && tnsList.length ? host.packTns(tnsList) : null, cce = cceList && cceList.length ? host.packCce(cceList) : null;
    const t0 = process.hrtime.bigint();
    let failed = null;
    try { g.engine.decodeBatch(host.packUnits(units), q, meta, pcm, tns, cce); } catch (err) { failed = err; }
    this.stats.engineNs += process.hrtime.bigint() - t0;
    this.stats.batches++; this.stats.units += units.length;
    if (!failed) {
        for (const p of parts) { p.dec.deliver(p, pcm, p.pcmBase, null); this.stats.frames += p.frames.length; }
        return;
    }
    /* the engine refused the batch as a whole (nothing was decoded, no state advanced): decoder by decoder, so that only the
     * stream whose frames it refuses pays for it */
    this.stats.retries++;
    for (const p of parts) {
        let refused = null;
        try { g.engine.decodeBatch(host.packUnits(p.units), q, meta, pcm, tns, cce); }
        catch (err) { refused = err instanceof Error ? err : new Error(String(err)); }
        p.dec.deliver(p, pcm, p.pcmBase, refused);
        if (!refused) this.stats.frames += p.frames.length;
    }
};

module.exports = { SharedEngine: SharedEngine };
