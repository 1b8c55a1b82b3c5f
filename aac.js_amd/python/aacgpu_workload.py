"""Synthetic AAC-LC workloads: unit tables, quantised spectra and band side info
for S streams x T consecutive frames, shaped like what the JavaScript host's parser emits.

Layout conventions used by bench.py and the tests:
  frame (s, t) -> frame index f = s * T + t
  pcm_offset   = f * 1024 * C            coef/meta block of channel c = f * C + c
"""
import os

import numpy as np

from aacgpu import UNIT_DTYPE

SWB_LONG_48 = np.concatenate([[0], np.cumsum([4] * 10 + [8] * 7 + [12] * 4 + [16] * 2 + [20] * 2 + [24] * 2 +
                                              [28] * 2 + [32] * 19 + [96])])
SWB_SHORT_48 = np.concatenate([[0], np.cumsum([4] * 5 + [8] * 3 + [12] * 3 + [16] * 3)])

SEQ_PATTERN_MIX = [0, 0, 1, 2, 2, 3, 0, 0]          # config 3: ONLY_LONG, LONG_START, EIGHT_SHORT x2, LONG_STOP, ...


def _band_of(offsets, n):
    b = np.zeros(n, np.int64)
    for i in range(len(offsets) - 1):
        b[offsets[i]:offsets[i + 1]] = i
    return b


def make_batch(n_streams, n_frames, layout=("cpe",), mix=False, seed=0xAAC00002, intensity=False, frame_base=0,
               stream_base=0):
    """Returns dict(units, q (int16 [blocks,1024]), meta (uint16 [blocks,120]), C, n_pcm, n_frames_total).

    mix=False: config 2 — all ONLY_LONG, KBD, maxSFB 49, common window, ms_used on even bands.
    mix=True : config 3 — per-stream sequence pattern [0,0,1,2,2,3,0,0] phase-shifted by stream, shorts grouped
               [3,4,1] with maxSFB 14, shapes alternating per frame.  frame_base continues the pattern across
               consecutive batches of the same streams.
    """
    rng = np.random.default_rng(seed + 7919 * frame_base)
    C = sum(2 if e == "cpe" else 1 for e in layout)
    F = n_streams * n_frames
    n_units = F * len(layout)
    units = np.zeros(n_units, UNIT_DTYPE)
    q = np.zeros((F * C, 1024), np.int16)
    meta = np.zeros((F * C, 120), np.uint16)

    s_idx = np.repeat(np.arange(n_streams), n_frames)
    t_idx = np.tile(np.arange(n_frames), n_streams) + frame_base
    if mix:
        pattern = [int(c) for c in os.environ.get("AACG_SEQ_PATTERN", "")] or SEQ_PATTERN_MIX      # diagnostic override, e.g. 22222222
        seq_f = np.array(pattern)[(t_idx + s_idx) % len(pattern)]
        shape_f = (t_idx & 1).astype(np.uint8)
    else:
        seq_f = np.zeros(F, np.int64)
        shape_f = np.ones(F, np.uint8)

    band_long, band_short = _band_of(SWB_LONG_48, 1024), _band_of(SWB_SHORT_48, 128)
    k = np.arange(1024)
    lam_long = 24.0 * np.exp(-k / 180.0)
    lam_short = np.tile(24.0 * np.exp(-(np.arange(128) * 8) / 180.0), 8)
    group_of_win = np.repeat([0, 1, 2], [3, 4, 1])

    ui = 0
    chan = 0
    for e_i, e in enumerate(layout):
        nc = 2 if e == "cpe" else 1
        sel = np.arange(F) * len(layout) + e_i
        u = units[sel]
        u["stream"] = s_idx + stream_base
        u["pcm_offset"] = np.arange(F, dtype=np.uint64) * 1024 * C
        u["channel"] = chan
        u["n_out_ch"] = C
        u["n_ch"] = nc
        u["flags"] = 3 if nc == 2 else 0          # common window + mask present
        u["coef_offset"] = np.arange(F) * C + chan
        u["meta_offset"] = np.arange(F) * C + chan
        is_short = seq_f == 2
        for c in range(nc):
            u["ch"]["window_sequence"][:, c] = seq_f
            u["ch"]["window_shape"][:, c] = shape_f
            u["ch"]["max_sfb"][:, c] = np.where(is_short, 14, 49)
            u["ch"]["group_count"][:, c] = np.where(is_short, 3, 1)
            gl = np.zeros((F, 8), np.uint8)
            gl[:, 0] = 1
            gl[is_short, :3] = [3, 4, 1]
            u["ch"]["group_len"][:, c] = gl
        units[sel] = u

        for c in range(nc):
            blocks = np.arange(F) * C + chan + c
            # two-sided geometric magnitudes with scale lambda(k)
            lam = np.where(is_short[:, None], lam_short[None, :], lam_long[None, :])
            mag = np.floor(rng.exponential(1.0, (F, 1024)) * lam * 0.5)
            sign = rng.integers(0, 2, (F, 1024)) * 2 - 1
            q[blocks] = np.clip(mag * sign, -8190, 8190).astype(np.int16)
            # band side info: spectral codebooks 1..11, scalefactor index 248 +- 8, ms_used on even bands (left)
            nb = np.where(is_short, 3 * 14, 49)
            bt = rng.integers(1, 12, (F, 120)).astype(np.uint16)
            sf = (248 + rng.integers(-8, 9, (F, 120))).astype(np.uint16)      # 2^(12 +- 2): PCM near -17 dBFS
            if intensity and c == 1:
                is_band = rng.random((F, 120)) < 0.15
                bt = np.where(is_band, rng.integers(14, 16, (F, 120)), bt).astype(np.uint16)
                sf = np.where(is_band, 200 + rng.integers(-20, 21, (F, 120)), sf).astype(np.uint16)
            m = sf | (bt << 12)
            if nc == 2 and c == 0:
                sfb = np.where(is_short[:, None], np.arange(120)[None, :] % 14, np.arange(120)[None, :])
                m = m | np.where(sfb % 2 == 0, 0x400, 0).astype(np.uint16)
            m = np.where(np.arange(120)[None, :] < nb[:, None], m, 0).astype(np.uint16)
            meta[blocks] = m
        chan += nc
        ui += 1
    return dict(units=units, q=q, meta=meta, C=C, n_pcm=F * 1024 * C, n_frames_total=F)


GROUPINGS = [[8], [3, 4, 1], [1] * 8, [2, 6], [4, 4], [1, 7], [2, 2, 2, 2], [1, 1, 6]]


def random_batch(seed, n_streams=3, max_frames=20):
    """Fuzz workload: per stream a random channel layout and frame count; per frame and element random window
    sequences (any order, also illegal transitions: the filterbank is defined for all of them), shapes, previous
    shapes, maxSFB, short-window groupings, common / split windows, band types (zero, codebooks, escape, intensity),
    MS masks (none / per band / all), scalefactors, and quantised values including escape-range magnitudes."""
    rng = np.random.default_rng(seed)
    band_long, band_short = _band_of(SWB_LONG_48, 1024), _band_of(SWB_SHORT_48, 128)
    units, qs, metas = [], [], []
    block = 0
    pcm = 0
    max_c = 0
    for s in range(n_streams):
        layout = [("cpe",), ("sce",), ("cpe", "sce"), ("sce", "cpe", "cpe", "sce"), ("cpe", "cpe", "cpe", "cpe")][rng.integers(0, 5)]
        C = sum(2 if e == "cpe" else 1 for e in layout)
        max_c = max(max_c, C)
        T = int(rng.integers(1, max_frames + 1))
        for t in range(T):
            chan = 0
            for e in layout:
                nc = 2 if e == "cpe" else 1
                u = np.zeros(1, UNIT_DTYPE)
                u["stream"] = s
                u["pcm_offset"] = pcm
                u["channel"] = chan
                u["n_out_ch"] = C
                u["n_ch"] = nc
                u["coef_offset"] = block
                u["meta_offset"] = block
                common = nc == 2 and rng.random() < 0.8
                mask_mode = int(rng.integers(0, 4)) if common else 0      # 0 none, 1/2 per band, 3 all
                u["flags"] = (1 if common else 0) | (2 if mask_mode else 0)
                infos = []
                for c in range(nc):
                    if c == 1 and common:
                        infos.append(infos[0])
                        continue
                    seq = int(rng.integers(0, 4))
                    gl = GROUPINGS[rng.integers(0, len(GROUPINGS))] if seq == 2 else [1]
                    infos.append(dict(seq=seq, shape=int(rng.integers(0, 2)), prev=int(rng.integers(0, 2)),
                                      max_sfb=int(rng.integers(0, 15)) if seq == 2 else int(rng.integers(0, 50)), gl=gl))
                for c, inf in enumerate(infos):
                    u["ch"]["window_sequence"][0, c] = inf["seq"]
                    u["ch"]["window_shape"][0, c] = inf["shape"]
                    u["ch"]["window_shape_prev"][0, c] = inf["prev"]
                    u["ch"]["max_sfb"][0, c] = inf["max_sfb"]
                    u["ch"]["group_count"][0, c] = len(inf["gl"])
                    u["ch"]["group_len"][0, c, :len(inf["gl"])] = inf["gl"]
                    nb = len(inf["gl"]) * inf["max_sfb"]
                    bt = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11], size=120).astype(np.uint16)
                    sfi = (232 + rng.integers(-12, 13, 120)).astype(np.uint16)
                    if c == 1:
                        isb = rng.random(120) < 0.2
                        bt = np.where(isb, rng.integers(14, 16, 120), bt).astype(np.uint16)
                        sfi = np.where(isb, 200 + rng.integers(-16, 17, 120), sfi).astype(np.uint16)
                    m = sfi | (bt << 12)
                    if c == 0 and mask_mode:
                        used = np.ones(120, bool) if mask_mode == 3 else rng.random(120) < 0.5
                        m = m | np.where(used, 0x400, 0).astype(np.uint16)
                    m = np.where(np.arange(120) < nb, m, rng.integers(0, 65536, 120) & 0xCFFF).astype(np.uint16)   # junk beyond nb, never NOISE
                    metas.append(m)
                    lam = np.tile(24.0 * np.exp(-(np.arange(128) * 8) / 180.0), 8) if inf["seq"] == 2 else 24.0 * np.exp(-np.arange(1024) / 180.0)
                    qq = np.floor(rng.exponential(1.0, 1024) * lam * 0.5) * (rng.integers(0, 2, 1024) * 2 - 1)
                    big = rng.random(1024) < 0.004
                    qq = np.where(big, rng.integers(-8190, 8191, 1024), qq)
                    qs.append(np.clip(qq, -8190, 8190).astype(np.int16))
                units.append(u)
                block += nc
                chan += nc
            pcm += 1024 * C
    return dict(units=np.concatenate(units), q=np.stack(qs), meta=np.stack(metas), n_pcm=pcm, n_streams=n_streams, max_channels=max_c)


# ---- TNS side info (AACG_TNS_SPEC) -------------------------------------------------------------------
def tns_coef_table(res_bits, compress):
    """The values a parser looks up for a `res_bits`-bit (3 or 4) TNS coefficient field, `compress` dropping the
    top bit (ISO/IEC 14496-3 4.6.9.3: sin(q / iqfac), with the sign convention of tns.js:49-63)."""
    n = 1 << (res_bits - compress)
    out = np.zeros(n, np.float32)
    iqfac = ((1 << (res_bits - 1)) - 0.5) / (np.pi / 2)
    iqfac_m = ((1 << (res_bits - 1)) + 0.5) / (np.pi / 2)
    for i in range(n):
        s = i - n if i >= n // 2 else i               # two's complement field
        out[i] = np.float32(-np.sin(s / (iqfac if s >= 0 else iqfac_m)))
    return out


def add_tns(batch, seed=0x7115, p_channel=0.6, max_order_long=12, max_order_short=7, wild=False):
    """Adds TNS side info to a batch from make_batch / random_batch: returns (units copy with CHAN_TNS_PRESENT
    and tns_offset set, TNS_DTYPE array).  Long windows get 1..3 filters, short windows 0..1 per window,
    lengths / orders / directions random, coefficients from the 3- and 4-bit tables.  The reflection
    coefficients decay with their index, as an encoder's do (|k_i| <= max(0.25, 0.98 * 0.75^i)); wild=True draws
    every one from the full table instead, which yields near-unstable filters with gains of 100 and more."""
    from aacgpu import TNS_DTYPE, CHAN_TNS_PRESENT
    rng = np.random.default_rng(seed)
    units = batch["units"].copy()
    recs = []
    for i in range(len(units)):
        u = units[i]
        n_ch = int(u["n_ch"])
        has = [rng.random() < p_channel for _ in range(n_ch)]
        if not any(has):
            continue
        units[i]["tns_offset"] = len(recs)             # n_ch consecutive records, unused ones stay empty
        for c in range(n_ch):
            rec = np.zeros((), TNS_DTYPE)
            if has[c]:
                short = int(u["ch"][c]["window_sequence"]) == 2
                res = int(rng.integers(3, 5))
                for w in range(8 if short else 1):
                    nf = int(rng.integers(0, 2)) if short else int(rng.integers(1, 4))
                    rec["n_filt"][w] = nf
                    for f in range(nf):
                        slot = w if short else f
                        order = int(rng.integers(0, (max_order_short if short else max_order_long) + 1))
                        tab = tns_coef_table(res, int(rng.integers(0, 2)))
                        rec["filt"][slot]["length"] = int(rng.integers(0, 16 if short else 50))
                        rec["filt"][slot]["order"] = order
                        rec["filt"][slot]["direction"] = int(rng.integers(0, 2))
                        for k in range(order):
                            ok = tab if wild else tab[np.abs(tab) <= max(0.25, 0.98 * 0.75 ** k)]
                            rec["filt"][slot]["coef"][k] = ok[rng.integers(0, len(ok))]
                units[i]["ch"][c]["flags"] |= CHAN_TNS_PRESENT
            recs.append(rec)
    tns = np.array(recs, TNS_DTYPE) if recs else np.zeros(1, TNS_DTYPE)
    return units, tns


# ---- perceptual noise substitution (AACG_PNS_SPEC) ----------------------------------------------------
def add_pns(batch, seed=0x9125, p_unit=0.7, p_band=0.25):
    """Turns random coded bands into NOISE_BT bands (band type 13, the scalefactor word = the band's energy
    scalefactor, here 2^((sf - 200) / 4) around the level of the other bands) and flags the units
    (UNIT_HAS_PNS).  Returns (units copy, meta copy); the quantised values of those bands are left as they are —
    a parser never delivers any, and the engine must not read them."""
    from aacgpu import UNIT_HAS_PNS
    rng = np.random.default_rng(seed)
    units, meta = batch["units"].copy(), batch["meta"].copy()
    for i in range(len(units)):
        if rng.random() >= p_unit:
            continue
        hit = False
        for c in range(int(units[i]["n_ch"])):
            ci = units[i]["ch"][c]
            nb = int(ci["group_count"]) * int(ci["max_sfb"])
            m = meta[int(units[i]["meta_offset"]) + c]
            for b in range(nb):
                bt = int(m[b]) >> 12
                if 1 <= bt <= 11 and rng.random() < p_band:
                    m[b] = (13 << 12) | (m[b] & 0x0400) | int(260 + rng.integers(-12, 13))     # keeps ms_used, drops negate
                    hit = True
        if hit:
            units[i]["flags"] |= UNIT_HAS_PNS
    return units, meta


def add_tns_config3(batch, seed=0xAAC00003):
    """TNS side info on every channel-frame — long windows one filter of order 12 over 20
    bands, EIGHT_SHORT one filter of order 7 per window (over all 14 bands here); directions alternate, reflection
    coefficients from the 4-bit table with the decaying envelope of add_tns."""
    from aacgpu import TNS_DTYPE, CHAN_TNS_PRESENT
    rng = np.random.default_rng(seed)
    units = batch["units"].copy()
    tab = tns_coef_table(4, 0)
    recs = []
    for i in range(len(units)):
        units[i]["tns_offset"] = len(recs)
        for c in range(int(units[i]["n_ch"])):
            rec = np.zeros((), TNS_DTYPE)
            short = int(units[i]["ch"][c]["window_sequence"]) == 2
            for w in range(8 if short else 1):
                rec["n_filt"][w] = 1
                f = rec["filt"][w]
                order = 7 if short else 12
                f["length"], f["order"], f["direction"] = (14 if short else 20), order, (i + c + w) & 1
                for k in range(order):
                    ok = tab[np.abs(tab) <= max(0.25, 0.98 * 0.75 ** k)]
                    f["coef"][k] = ok[rng.integers(0, len(ok))]
            units[i]["ch"][c]["flags"] |= CHAN_TNS_PRESENT
            recs.append(rec)
    return units, np.array(recs, TNS_DTYPE)


def add_cce(batch, points=(2,), seed=0xAAC00005, max_targets=4):
    """Coupling channel elements (AACG_CCE_SPEC) for a make_batch() workload: per frame one CCE per entry of `points`
    (0 before TNS, 1 after TNS: dependent, per-band gains; 2 after the IMDCT: independent, one gain per target), with the
    frame's window info, a quantised spectrum of its own (blocks appended behind the batch's), 1..max_targets target
    channels and gains +-2^(-t/4).  CCE number k of a frame lives at stream channel C + k; its unit sits at a varying
    position among the frame's elements (the order of elements in a raw_data_block is free).
    Returns (units, q, meta, cce records); the engine needs max_channels = C + len(points)."""
    from aacgpu import CCE_DTYPE, UNIT_CCE
    rng = np.random.default_rng(seed)
    units, C = batch["units"], batch["C"]
    F = batch["n_frames_total"]
    L = len(units) // F
    K = len(points)
    blocks0 = batch["q"].shape[0]
    q = np.concatenate([batch["q"], np.zeros((F * K, 1024), np.int16)])
    meta = np.concatenate([batch["meta"], np.zeros((F * K, 120), np.uint16)])
    cce = np.zeros(F * K, CCE_DTYPE)
    out = np.zeros(F * (L + K), UNIT_DTYPE)
    k1024 = np.arange(1024)
    for f in range(F):
        frame = units[f * L:(f + 1) * L]
        extra = np.zeros(K, UNIT_DTYPE)
        for k, point in enumerate(points):
            idx, blk = f * K + k, blocks0 + f * K + k
            u = extra[k]
            u["stream"], u["pcm_offset"], u["n_out_ch"] = frame[0]["stream"], frame[0]["pcm_offset"], C
            u["channel"], u["n_ch"], u["flags"], u["reserved1"] = C + k, 1, UNIT_CCE, idx
            u["coef_offset"] = u["meta_offset"] = blk
            u["ch"][0] = frame[0]["ch"][0]
            u["ch"][0]["flags"] = 0
            short = int(u["ch"][0]["window_sequence"]) == 2
            lam = np.tile(16.0 * np.exp(-(np.arange(128) * 8) / 200.0), 8) if short else 16.0 * np.exp(-k1024 / 200.0)
            q[blk] = (np.floor(rng.exponential(1.0, 1024) * lam * 0.5) * (rng.integers(0, 2, 1024) * 2 - 1)).astype(np.int16)
            nb = int(u["ch"][0]["group_count"]) * int(u["ch"][0]["max_sfb"])
            bt = rng.integers(0, 12, 120).astype(np.uint16)                   # type 0 = ZERO_BT: skipped by the coupling
            meta[blk, :nb] = ((244 + rng.integers(-6, 7, 120)).astype(np.uint16) | (bt << 12))[:nb]
            rec = cce[idx]
            rec["coupling_point"] = point
            targets = rng.choice(C, size=int(rng.integers(1, min(max_targets, C) + 1)), replace=False)
            rec["n_targets"] = len(targets)
            for t, ch in enumerate(targets):
                rec["target"][t]["channel"], rec["target"][t]["gain_list"] = ch, t
                rec["gain"][t] = (rng.integers(0, 2, 120) * 2 - 1) * 2.0 ** (-rng.integers(0, 12, 120) / 4.0)
        at = f % (L + 1)
        out[f * (L + K):(f + 1) * (L + K)] = np.concatenate([frame[:at], extra, frame[at:]])
    return out, q, meta, cce


# ---- device front end: stand-in codebooks and a hand-assembled frame (smoke test, no Node needed) -------------------
def standin_codebooks():
    """The 12 codebooks' alphabets (ISO/IEC 14496-3 4.A.1) with the simplest complete prefix code over each: the first
    2^(k+1) - n symbols get k bits, the rest k + 1 (k = floor(log2 n)), canonical assignment.  Not the standard's code
    words (those are not in this repository); a parser does not care.  Returns (entries, counts)."""
    import itertools
    from aacgpu import CODE_ENTRY_DTYPE
    books = [[(i,) for i in range(121)]]
    for lo, hi, dim in ((-1, 1, 4), (-1, 1, 4), (0, 2, 4), (0, 2, 4), (-4, 4, 2), (-4, 4, 2), (0, 7, 2), (0, 7, 2), (0, 12, 2), (0, 12, 2), (0, 16, 2)):
        books.append(list(itertools.product(range(lo, hi + 1), repeat=dim)))
    entries = np.zeros(sum(len(b) for b in books), CODE_ENTRY_DTYPE)
    counts = np.array([len(b) for b in books], np.uint32)
    at = 0
    for b in books:
        n = len(b)
        k = n.bit_length() - 1
        short = (2 << k) - n if n != (1 << k) else n
        code, prev = 0, k
        for i, v in enumerate(b):
            length = k if i < short else k + 1
            code <<= length - prev
            prev = length
            entries[at]["code"], entries[at]["len"] = code, length
            entries[at]["v"][:len(v)] = v
            code += 1
            at += 1
    return entries, counts


def tiny_frame(entries, counts, pairs=((1, -2), (0, 3), (-4, 4), (0, 0))):
    """One raw_data_block: an SCE, ONLY_LONG, two scalefactor bands coded with book 5 (signed pairs), scalefactor steps 0.
    Returns (bytes, expected first coefficients, expected band word)."""
    start = [0] + [int(x) for x in np.cumsum(counts)]
    def code(book, values):
        for e in entries[start[book]:start[book + 1]]:
            if tuple(e["v"][:len(values)]) == tuple(values):
                return int(e["code"]), int(e["len"])
        raise KeyError(values)
    bits = []
    def put(v, n):
        bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))
    put(0, 3); put(0, 4)                                  # SCE, tag 0
    put(100, 8)                                           # global_gain
    put(0, 1); put(0, 2); put(0, 1); put(2, 6); put(0, 1)  # ics_info: ONLY_LONG, sine, max_sfb 2, no prediction
    put(5, 4); put(2, 5)                                  # one section: book 5 over both bands
    for _ in range(2):
        put(*code(0, (60,)))                              # scalefactor step 0: 100 -> table index 200
    put(0, 1); put(0, 1); put(0, 1)                       # no pulse, TNS, gain control
    for p in pairs:
        put(*code(5, p))
    put(7, 3)                                             # END
    while len(bits) % 8:
        bits.append(0)
    data = np.packbits(np.array(bits, np.uint8))
    return data, np.array([v for p in pairs for v in p], np.int16), (5 << 12) | 200
