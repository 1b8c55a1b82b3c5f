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
 * raised BEFORE anything was launched (a refused unit, a capacity or layout error) is retried decoder by decoder, so that it
 * only costs the stream that caused it; a device error goes to every stream of the batch (their state may have moved).
 *
 * { resident: true } — the front end on the device as well: the decoders do not parse at all.  Each keeps the bytes its
 * demuxer fed it; a flush finds the next frames' boundaries (the 13-bit ADTS frame_length: six byte reads per frame), hands the
 * bytes of ALL streams to ONE native call (aacg_pipeline_decode: bit streams up, parse kernel, kept plan refreshed on the device,
 * transform kernel, PCM down) and slices what comes back.  JavaScript touches no coefficient and no unit record; what
 * readChunk() returns are views of the batch's PCM array.  For ADTS byte streams of any channel configuration 1..8 (one
 * pipeline per sample rate and channel count; a stream's element layout — SCE + CPE + CPE + LFE, say — is learnt on the device
 * from its first frame, coupling elements are parsed and dropped), as the reference executes them (TNS identity, no PNS /
 * coupling modes).  MP4 samples ('mp4a': packets without ADTS headers, several to a buffer, whose boundaries only a parse
 * finds) and decoders with spec modes take the parsing route above on the same SharedEngine.
 * Replaces, per batch, what src/decoder.js:125-216 does per frame.
 */
'use strict';
const host = require('./index.js');

function SharedEngine(opts) {
    opts = opts || {};
    this.opts = opts;
    this.maxStreams = opts.maxStreams || 256;
    this.maxChannels = opts.maxChannels || 8;
    this.groups = new Map();                              // sampleIndex -> { engine, decoders: [], free: [] }
    this.resident = !!opts.resident;                      // bytes -> PCM in one native call per flush (aacg_pipeline_*)
    this.lookahead = opts.lookahead || 16;                // resident: frames per stream and flush
    /* resident: where a flush's PCM lives.  0 (default): memory of its own, the caller's for as long as it keeps any frame of the
     * flush (what the reference's readChunk() promises: a fresh array per frame) — a page-locked allocation per flush.  K > 0: K
     * page-locked buffers made once and used in turn: a frame is valid until its SharedEngine has flushed K more times, i.e. for
     * at least (K - 1) x lookahead further frames of its stream; a consumer that keeps frames longer copies them (frame.slice()) */
    this.pcmRing = opts.pcmRing | 0;
    /* resident: decode the NEXT flush's batch while the caller consumes this one (addon.pipelineSubmit / pipelineCollect:
     * aacg_pipeline_submit enqueues the batch on one of the pipeline's lanes and returns; no thread of the addon's is involved).
     * A flush then collects the batch submitted by the flush before and submits the one after; streams are read one flush
     * (lookahead frames) further ahead than they are consumed, and with pcmRing a frame is valid for one flush less.  On by
     * default since round 6 ({ overlap: false } for a flush that decodes what it returns). */
    this.overlap = opts.overlap === undefined ? this.resident : !!opts.overlap;
    this.stats = { batches: 0, frames: 0, units: 0, engineNs: 0n, retries: 0 };
}

/* error codes the engine raises before it has launched anything (include/aacgpu.h): a batch refused with one of these has
 * advanced no stream, and may be retried in parts */
const BEFORE_LAUNCH = [-1, -4, -5, -6];                   // INVALID_ARG, CAPACITY, UNSUPPORTED, LAYOUT_CHANGE
function refusedBeforeLaunch(err) {
    const m = /failed \((-?\d+)\)/.exec(String(err && err.message));
    return !!m && BEFORE_LAUNCH.indexOf(parseInt(m[1], 10)) >= 0;
}

/* called by GpuAACDecoder.setCookie: the engine for the decoder's sample rate and a stream slot in it */
SharedEngine.prototype.attach = function (dec) {
    const cfg = dec.config;
    if (cfg.chanConfig + dec.maxCoupling > this.maxChannels) throw new Error('SharedEngine: ' + cfg.chanConfig + ' channels exceed maxChannels ' + this.maxChannels);
    if (dec.group) this.detach(dec);                       // a new cookie: a new stream
    /* the resident route reads ADTS frame lengths itself: an 'mp4a' track's samples have none (aurora.js feeds them through
     * feedPacket, which the device route never sees — ADVICE round 5: such a decoder used to deliver nothing, without an error) */
    const adts = !dec.format || dec.format.formatID === undefined || dec.format.formatID === 'aac ';
    if (this.resident && adts && cfg.profile === 2 && cfg.chanConfig >= 1 && cfg.chanConfig <= 8 && !dec.tnsMode && !dec.pnsMode && !dec.cceMode && !dec.carryWindowShape)
        return this.attachResident(dec);
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

/* resident route: one pipeline (engine + parser on the device) per (sample rate, channel count) */
SharedEngine.prototype.attachResident = function (dec) {
    const cfg = dec.config, key = 'r' + cfg.sampleIndex + '/' + cfg.chanConfig;
    let g = this.groups.get(key);
    if (!g) {
        const addon = host.loadAddon(), rec = require('./codebooks.js').load(this.opts).toEntryRecords();
        const outI16 = (this.opts.outputKind | 0) === host.OUTPUT_I16;
        const pipeline = addon.pipelineCreate({ deviceOrdinal: this.opts.deviceOrdinal | 0, sampleIndex: cfg.sampleIndex, maxStreams: this.maxStreams,
                                                channels: cfg.chanConfig, maxFrames: this.lookahead, outputKind: this.opts.outputKind | 0,
                                                parseOptions: (this.opts.applyPulses ? 1 : 0) | (this.opts.referenceQuirks !== false ? 2 : 0) }, rec.entries, rec.counts);
        g = { resident: true, addon: addon, pipeline: pipeline, channels: cfg.chanConfig, outI16: outI16, decoders: [], free: [], next: 0 };
        this.groups.set(key, g);
    }
    const slot = g.free.length ? g.free.pop() : g.next++;
    if (slot >= this.maxStreams) { g.next--; throw new Error('SharedEngine: more than ' + this.maxStreams + ' streams at one sample rate'); }
    dec.stream = slot; dec.group = g; dec.resident = true;
    dec.rbuf = new Uint8Array(0); dec.rpos = 0;
    dec.engine = { resetStream: function (s) { g.addon.pipelineResetStream(g.pipeline, s); } };
    g.decoders.push(dec);
};

SharedEngine.prototype.detach = function (dec) {
    const g = dec.group;
    if (!g) return;
    g.decoders.splice(g.decoders.indexOf(dec), 1);
    g.free.push(dec.stream);
    dec.group = null;
};

/* how many complete ADTS frames (at most `max`) start at byte `at` of `b`: their lengths into `lens`; -1 where the bytes at a
 * frame boundary are not an ADTS header (the reference's 'Invalid ADTS header.', adts_demuxer.js:29) — or are one whose
 * frame_length is shorter than the header itself: a corrupt length field must not stall the stream silently (ADVICE round 5) */
function scanFrames(b, at, max, lens) {
    let n = 0;
    while (n < max && at + 7 <= b.length) {
        if (b[at] !== 0xff || (b[at + 1] & 0xf0) !== 0xf0) return n ? n : -1;
        const len = ((b[at + 3] & 3) << 11) | (b[at + 4] << 3) | (b[at + 5] >> 5);
        if (len < 7) return n ? n : -1;
        if (at + len > b.length) break;
        lens[n++] = len; at += len;
    }
    return n;
}

/* resident flush: the next frames of every stream that has run dry or has room, ONE native call, views of its PCM handed out */
SharedEngine.prototype.prepareResident = function (g, room) {
    const L = this.lookahead;
    const part = [];
    let F = L;
    for (const dec of g.decoders) {
        if (dec.queue.length >= room) continue;            // a slow reader's queue does not grow with every flush of its peers
        if (!dec.rlens) dec.rlens = new Uint32Array(L);
        const n = scanFrames(dec.rbuf, dec.rpos, L, dec.rlens);
        if (n < 0) { if (!dec.badHeader) { dec.badHeader = true; dec.queue.push(new Error('Invalid ADTS header.')); } continue; }
        if (n === 0) continue;
        part.push(dec); F = Math.min(F, n);
    }
    if (!part.length) return null;
    const S = part.length, frames = new Uint32Array(2 * S * F), slots = new Uint32Array(S), starts = new Uint32Array(S + 1);
    let total = 0;
    for (let s = 0; s < S; s++) {
        const lens = part[s].rlens;
        starts[s] = total; slots[s] = part[s].stream;
        for (let f = 0, i = 2 * s * F; f < F; f++, i += 2) { frames[i] = total; frames[i + 1] = lens[f]; total += lens[f]; }
    }
    starts[S] = total;
    const bytes = new Uint8Array(total);
    for (let s = 0; s < S; s++) {
        const dec = part[s], len = starts[s + 1] - starts[s];
        bytes.set(dec.rbuf.subarray(dec.rpos, dec.rpos + len), starts[s]);
        dec.rpos += len;
    }
    return { part: part, S: S, F: F, frames: frames, slots: slots, bytes: bytes, results: new Uint8Array(8 * S * F) };
};

/* the batch's PCM: one array on page-locked memory the device wrote into; a frame is a view of it (the memory returns to the
 * addon's pool when the last of the batch's frames has been collected) */
SharedEngine.prototype.deliverResident = function (g, b, out, failed) {
    const C = g.channels, part = b.part, S = b.S, F = b.F, results = b.results;
    this.stats.batches++; this.stats.units += S * F;
    if (failed) { for (const dec of part) dec.queue.push(failed); return; }
    const pcm = out.pcm, refused = out.refused, per = 1024 * C;
    for (let s = 0; s < S; s++) {
        const q = part[s].queue;
        for (let f = 0, i = s * F; f < F; f++, i++) {
            /* a frame the device refused: the reference's message for its status where the frame is reached (it was decoded as
             * silence: the stream goes on); a frame whose elements are not the ones its stream began with has a status of its own
             * (AACG_PARSE_LAYOUT, set where the plan's records are refreshed) */
            if (refused && results[8 * i]) q.push(new Error(g.addon.parseStatusString(results[8 * i])));
            else q.push(pcm.subarray(i * per, (i + 1) * per));
        }
    }
    this.stats.frames += S * F;
};

SharedEngine.prototype.flushResident = function (g) {
    const L = this.lookahead, C = g.channels, ringElems = this.pcmRing ? this.maxStreams * L * 1024 * C : 0;
    const args = (b) => [g.pipeline, b.bytes, b.frames, b.slots, b.F, b.results, C, this.pcmRing, ringElems];
    const t0 = process.hrtime.bigint();
    let b = null, out = null, failed = null;
    if (this.overlap && g.pending) {                       // the batch the flush before this one submitted
        b = g.pending; g.pending = null;
        try { out = g.addon.pipelineCollect(g.pipeline); } catch (err) { failed = err instanceof Error ? err : new Error(String(err)); }
    } else {
        b = this.prepareResident(g, L);
        if (!b) return;
        try { out = g.addon.pipelineDecode.apply(null, args(b)); } catch (err) { failed = err instanceof Error ? err : new Error(String(err)); }
    }
    this.stats.engineNs += process.hrtime.bigint() - t0;
    this.deliverResident(g, b, out, failed);
    if (this.overlap && !failed) {                         // ... and the one after it: decoded while the caller reads this one
        const nb = this.prepareResident(g, 2 * L);
        if (nb) {
            try { g.addon.pipelineSubmit.apply(null, args(nb)); g.pending = nb; }
            catch (err) { this.deliverResident(g, nb, null, err instanceof Error ? err : new Error(String(err))); }
        }
    }
};

/* every registered decoder parses ahead what it has buffered; one batch per engine */
SharedEngine.prototype.flush = function () {
    for (const g of this.groups.values()) this.flushGroup(g);
};

SharedEngine.prototype.flushGroup = function (g) {
    if (g.resident) return this.flushResident(g);
    const tnsList = (this.opts.tnsMode | 0) === host.TNS_SPEC ? [] : null, cceList = (this.opts.cceMode | 0) === host.CCE_SPEC ? [] : null;
    const parts = [];
    let block = 0, pcmAt = 0;
    for (const dec of g.decoders) {
        let part;
        if (dec.failedProfile || dec.queue.length >= dec.lookahead) continue;     // a paused reader's queue does not grow with every flush of its peers
        try { part = dec.collectAhead(block, pcmAt, tnsList, cceList); }
        catch (err) { dec.failedProfile = true; dec.queue.push(err instanceof Error ? err : new Error(String(err))); continue; }    // unsupported profile: that decoder's own error, once
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
    /* a device error: the launch may have advanced every stream's state — decoding the same frames again would advance it
     * twice and deliver wrong PCM without an error: the error goes to every stream of the batch instead */
    if (!refusedBeforeLaunch(failed)) {
        for (const p of parts) p.dec.deliver(p, pcm, p.pcmBase, failed);
        return;
    }
    /* the engine refused the batch as a whole before launching anything (nothing was decoded, no state advanced): decoder by
     * decoder, so that only the stream whose frames it refuses pays for it */
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
