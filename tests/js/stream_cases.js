/*
 * tests/js/stream_cases.js — TEST KIT: the stream layouts and the random-frame generator shared by
 * test_frontend.js (JavaScript front end) and parse_cases.js (inputs / expected outputs for the device front end).
 */
'use strict';

function layoutChannels(layout) { return layout.reduce(function (a, e) { return a + (e === 'cpe' ? 2 : e === 'sce' || e === 'lfe' ? 1 : 0); }, 0); }

/* one random frame for `layout` (element kinds, e.g. ['cpe'] or ['sce','dse','cpe','fil','cce','lfe']) */
function randomFrame(wr, rng, layout, seqOf, o) {
    const elements = [];
    layout.forEach(function (kind, ei) {
        const id = rng.below(16);
        if (kind === 'sce' || kind === 'lfe') {
            elements.push({ type: kind, id: id, ch: [wr.randomChannel(rng, { seq: seqOf(ei), tns: o.tns && rng.chance(0.5), pulse: o.pulse && rng.chance(0.5), allowPNS: o.pns })] });
        } else if (kind === 'cpe') {
            const common = !o.noCommon && rng.below(5) !== 0;
            const L = wr.randomChannel(rng, { seq: seqOf(ei), tns: o.tns && rng.chance(0.5), pulse: o.pulse && rng.chance(0.4), allowPNS: o.pns });
            const R = wr.randomChannel(rng, { seq: seqOf(ei + 3), template: common ? L : null, allowIS: true, tns: o.tns && rng.chance(0.5), pulse: o.pulse && rng.chance(0.4), allowPNS: o.pns });
            const e = { type: 'cpe', id: id, commonWindow: common, mask: common ? rng.below(3) : 0, ch: [L, R] };
            if (e.mask === 1) { e.msUsed = []; for (let i = 0; i < L.groupLen.length * L.maxSFB; i++) e.msUsed.push(rng.below(2)); }
            elements.push(e);
        } else if (kind === 'dse') {
            const bytes = [];
            for (let i = rng.chance(0.2) ? 255 + rng.below(40) : rng.below(40); i > 0; i--) bytes.push(rng.below(256));
            elements.push({ type: 'dse', id: id, align: rng.below(2) === 1, bytes: bytes });
        } else if (kind === 'fil') {
            elements.push({ type: 'fil', count: rng.chance(0.3) ? 15 + rng.below(60) : rng.below(15) });
        } else if (kind === 'cce') {
            const targets = [], lists = [];
            for (let i = 1 + rng.below(3); i > 0; i--) {
                const t = { pair: rng.below(2) === 1, id: rng.below(16), sel: rng.below(4) };
                targets.push(t); lists.push(0);
                if (t.pair && t.sel === 3) lists.push(0);
            }
            lists.shift();                                             // the first gain list is implicit
            const ch = wr.randomChannel(rng, { seq: seqOf(ei) });
            if (o.cceNoZero) for (let i = 0; i < ch.bandTypes.length; i++) if (ch.bandTypes[i] === 0) { ch.bandTypes[i] = 11; ch.sf[i] = ch.globalGain; ch.split[i] = 1; }
            if (o.cceNoZero) fixScalefactors(ch);
            elements.push({ type: 'cce', id: id, point: o.ccePoint === undefined ? rng.below(4) : o.ccePoint, targets: targets, sign: rng.below(2), scale: rng.below(4), ch: [ch],
                            quirks: o.quirks !== false,
                            lists: lists.map(function () { return { cge: rng.below(2), common: rng.below(21) - 10, steps: [rng.below(9) - 4, rng.below(9) - 4, 0] }; }) });
        }
    });
    return elements;
}
/* after band types were edited: make the spectrum-class scalefactors a valid differential chain again */
function fixScalefactors(ch) {
    let spec = ch.globalGain;
    for (let i = 0; i < ch.bandTypes.length; i++)
        if (ch.bandTypes[i] >= 1 && ch.bandTypes[i] <= 11) { if (Math.abs(ch.sf[i] - spec) > 60) ch.sf[i] = spec; spec = ch.sf[i]; }
}

const PATTERN = [0, 0, 1, 2, 2, 3, 0, 1, 2, 3, 0, 0, 1, 3, 2, 0, 3, 1];
const CASES = [
    { name: 'stereo48', si: 3, layout: ['cpe'], frames: 18, o: { tns: true } },
    { name: 'mono44', si: 4, layout: ['sce'], frames: 9, o: { tns: true, pulse: true, pns: true } },
    { name: 'split16', si: 8, layout: ['cpe'], frames: 8, o: { noCommon: true, pulse: true } },
    { name: 'five1_48', si: 3, layout: ['sce', 'cpe', 'dse', 'cpe', 'fil', 'lfe'], frames: 6, o: { tns: true, pns: true } },
    { name: 'surround48', si: 3, layout: ['sce', 'cpe', 'dse', 'cpe', 'fil', 'lfe'], frames: 5, o: { tns: true } },
    { name: 'mono22', si: 7, layout: ['sce'], frames: 7, o: {} },
    { name: 'extras8k', si: 11, layout: ['fil', 'dse', 'cpe', 'cce', 'dse', 'fil'], frames: 6, o: { cceNoZero: true, ccePoint: 1 } },
    { name: 'cce96', si: 0, layout: ['cce', 'sce', 'cce', 'cpe'], frames: 6, o: {} },
    { name: 'cce_spec', si: 5, layout: ['cce', 'cpe', 'cce'], frames: 5, o: { quirks: false } },
];

module.exports = { randomFrame, fixScalefactors, layoutChannels, CASES, PATTERN };
