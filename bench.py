#!/usr/bin/env python3
"""bench.py — AAC-LC 48 kHz stereo frames/sec of the hot path on MI355X (BASELINE.json metric).

One step = one pass of the hot path (dequant + MS/IS -> IMDCT -> window -> overlap-add -> interleave)
over one batch of BASELINE config 2: 256 streams x 16 consecutive frames = 4096 stereo ONLY_LONG
frames (KBD, maxSFB 49, common window, ms_used on even bands), int16 quantised spectra + band side
info resident in HBM, float PCM written to HBM.  Consecutive steps are consecutive batches of the
same 256 streams (overlap state carried in the engine), rotating through NBUF distinct input/output
buffer sets so that no step is served from the 256 MiB Infinity Cache.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--input quant|spec] [--workload cfg2|cfg3|cfg4|cfg5] [--tns reference|spec]

Launches: consecutive steps go through aacg_decode_pipelined — the engine's internal HIP streams taken in turn, so that
step k + 1 starts on the compute units step k has already left; the chains of the two launches meet in rendezvous cells
(include/aacgpu.h; --serial: aacg_decode_device on one stream, every launch behind the one before it, the method of rounds
1-4).  ONE plan, ONE set of 256 streams continued launch after launch either way (`config.pipelines` 1).

Timing: W untimed warm-up steps, then the timed region of exactly K steps — R times back to back (--repeats, default 25;
regions shorter than 80 steps are repeated until about 2000 launches are timed, and timed in runs of consecutive repeats of at
least 100 launches — the marks go behind every G-th repeat, G x K >= 100: region_stats says why the median wants such runs, and
marks as dense as three per twenty launches cost the overlapped route up to 1.5 us per launch).  A timed run ends when its last launches —
one per pipeline stream: they run side by side — are complete: HIP events bound to those dispatches' completion
(aacg_decode_pipelined_timed; no marker packet enters a queue, nothing is joined inside the timed region); the opening mark is
recorded on the timing stream joined behind the warm-up steps; the whole set between barrier + synchronize on both sides
(--serial: every repeat between its own events on the launch stream).  `ms_per_step` = MEDIAN over the R repeats of (MAX over ranks of the repeat's event time) / K (SURVEY.md 8d asks
for a median; one 0.25 ms window says nothing about its own spread), `timing` carries min / max / first / R, and
`value` = frames of all ranks per step / that median.  The host's clock between the barriers is reported beside it
(`wall_ms_per_step`, `value_wall`: adds the launch latency of the first step and the wake-up after the last one).
Before the warm-up the GPU is loaded for --precondition-ms (default 300 ms, untimed, reported as
config.preconditioning): a few hundred steps are over before the clocks have ramped, and the same kernel then measures
12 % slower.

After the timed region (untimed): every stream is reset, one more step runs, and its PCM is compared with the
oracle on the whole batch (`parity_rms`, `parity_rel`; gate 1e-5 / 5e-6 — the run fails if it is missed).

N > 1: one rank per GPU under torch.distributed.run (the driver's launch line); started without it, `--gpus N`
launches the N ranks itself as a child process before touching the GPU.  Streams are sharded over ranks (every
rank decodes its own 256 streams: weak scaling, no data-path collective; RCCL only carries the barrier and the
max-over-ranks of the elapsed time).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak (6.29 TB/s measured copy)
STREAMS, FRAMES = 256, 16      # config 2: 4096 stereo frames per batch per GPU


def workload_shape(workload):
    """(mix, streams per GPU, frames per stream per step, element layout, channels) of a --workload.
    cfg2 (the metric's configuration) / cfg3: 256 streams x 16 frames; cfg4: 32 streams x 128 frames per GPU (BASELINE config 4:
    256 streams over 8 GPUs); cfg5: 3 CPE + LFE = 7 channels per frame."""
    mix = workload in ("cfg3", "cfg4", "cfg5")
    n_streams, n_frames = (32, 128) if workload == "cfg4" else (STREAMS, FRAMES)
    layout = ("cpe", "cpe", "cpe", "sce") if workload == "cfg5" else ("cpe",)
    return mix, n_streams, n_frames, layout, (7 if workload == "cfg5" else 2)


def region_stats(region_ms, steps, group=1):
    """Per-step time of the R timed repeats (each K steps): median (the figure quoted), min, max, first.
    group > 1: the median is taken over runs of `group` consecutive repeats.  With overlapped launches a repeat ends at the
    completion of launches that run beside the next repeat's first ones: its end is a maximum over those launches, a short
    repeat's time scatters by a launch either way and skews (at K = 20: 8.9-13.8 us per step around a mean of 11.6, median 0.1
    above it); the sum over consecutive repeats telescopes, so longer runs of repeats have the same mean and a fraction of the
    scatter.  min / max / each stay per repeat."""
    each = [float(t) / steps for t in region_ms]
    n = len(each)
    g = max(1, min(int(group), n))
    runs = [sum(each[i:i + g]) / g for i in range(0, n - g + 1, g)] if g > 1 else list(each)
    per = sorted(runs)
    m = len(per)
    med = per[m // 2] if m % 2 else 0.5 * (per[m // 2 - 1] + per[m // 2])
    out = {"repeats": n, "ms_per_step_median": med, "ms_per_step_min": min(each), "ms_per_step_max": max(each),
           "ms_per_step_first": each[0], "ms_per_step_each": each, "ms_per_step_mean": sum(each) / n}
    if g > 1:
        out["median_over"] = "%d runs of %d consecutive repeats (%d launches each)" % (m, g, g * steps)
    return out


def whole_job_value(world, frames_per_step, ms_per_step):
    """BASELINE metric: frames of ALL ranks per step / the step time (MAX over ranks already taken)."""
    return world * frames_per_step / (ms_per_step * 1e-3)


def backend_fields(requested, dist):
    """Top-level facts about the harness's process group: which backend actually carries the barrier / MAX, the world size IT
    reports, and whether RCCL was asked for and something else answered (a line with `collectives_backend_fallback` true must be
    read as "RCCL did not see these ranks", whatever its numbers say)."""
    if dist is None:
        return {"dist_backend": None, "dist_backend_requested": requested, "dist_world_size": 1, "collectives_backend_fallback": False}
    used = dist.get_backend()
    return {"dist_backend": used, "dist_backend_requested": requested, "dist_world_size": dist.get_world_size(),
            "collectives_backend_fallback": bool(requested == "nccl" and used != "nccl")}


def algorithmic_bytes_per_channel_frame(kind, chain_frames, pcm="f32"):
    """per channel-frame 4096 B spectrum (f32) or 2048 B coefficients + 240 B band side
    info (int16 path), 4096 B PCM out (2048 B as int16), + 8192/T B of overlap state (read + written once per chain
    of T consecutive frames: 16 for configs 2, 3, 5, 128 for config 4)."""
    return (4096 if kind == "spec" else 2048 + 240) + (2048 if pcm == "i16" else 4096) + 8192.0 / chain_frames


def measured_traffic(kernel):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate rocprofv3
    --pmc passes of this same command, tools/prof.sh); bench.py cannot run the profiler on itself, so the
    committed summary of the latest round under profiles/ is quoted — for the kernel that was PROFILED there and no
    other: None unless this run's kernel is that one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        for rec in d.values():
            if isinstance(rec, dict) and rec.get("kernel") == kernel and rec.get("workload", "cfg2") == "cfg2":
                return float(rec["traffic_bytes"]), os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError, OSError):
        pass
    return None, None


def cpu_baseline(kind, mix, layout, n_chan, budget_s=10.0, all_cores=True):
    """The oracle (plain C restatement of the reference algorithm, bit-exact with aac.js) on the host cores, on a
    bounded sample of the same workload: batches of 4 streams x 16 frames for ~budget_s on ONE core, then one such
    stream set per thread on ALL the cores this process may use for ~budget_s / 2 (oracle/orc_bench.c)."""
    import numpy as np
    import aacgpu_workload
    import orc
    o = orc.load()
    wl = aacgpu_workload.make_batch(n_streams=4, n_frames=FRAMES, mix=mix, layout=layout, seed=0xAAC00002)
    ov = np.zeros((4, n_chan, 1024), np.float32)
    coeffs = wl["q"]
    if kind == "spec":
        _, coeffs = o.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], ov, want_spec=True)
    meta = wl["meta"] if kind == "quant" else None
    o.decode_batch(wl["units"], coeffs, meta, wl["n_pcm"], ov)          # warm
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    per_batch = wl["n_frames_total"]
    n1, dt1 = o.bench_threads(1, budget_s, wl["units"], coeffs, meta, wl["n_pcm"], 4, n_chan)
    what = "4 streams x 16-frame batches, same generator as the GPU workload; oracle/aac_oracle.c, gcc -O2 -ffp-contract=off"
    single = {"value": n1 * per_batch / dt1, "unit": "frames/s", "cores": 1,
              "sample": "%d frames in %.1f s on 1 thread; %s" % (n1 * per_batch, dt1, what)}
    if not all_cores:
        # N > 1: the other ranks' host threads are parked in the closing barrier on the same cores: one thread only, said so
        return dict(single, kind="port", single_core=dict(single),
                    note="rank 0 of a multi-GPU run: one host thread (the all-cores figure is taken at N = 1, where no other rank shares the host)")
    nall, dtall = o.bench_threads(cores, budget_s / 2, wl["units"], coeffs, meta, wl["n_pcm"], 4, n_chan)
    out = {"value": nall * per_batch / dtall, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d frames in %.1f s on %d threads (one stream set each) of %d host cores; %s" % (nall * per_batch, dtall, cores, os.cpu_count(), what),
           "single_core": single}
    # the same path as plain JavaScript under Node, the stand-in for "aac.js's own Node path" (the reference cannot
    # travel to the GPU box); in the build container the real aac.js ran process()+interleave at 0.55x this port's rate
    # 
    import shutil
    import subprocess
    node = shutil.which("node")
    if node and not mix and kind == "quant" and n_chan == 2:
        try:
            r = subprocess.run([node, os.path.join(ROOT, "oracle", "js", "aac_port.js"), "bench", "4"],
                               capture_output=True, text=True, timeout=60)
            js = json.loads(r.stdout.strip().splitlines()[-1])
            out["js_port"] = {"value": js["frames_per_s"], "unit": "frames/s", "cores": 1, "node": js["node"],
                              "sample": "%d frames in %.1f s, oracle/js/aac_port.js" % (js["frames"], js["seconds"])}
        except Exception as exc:                        # baseline extra only; never fails the bench
            out["js_port"] = {"error": str(exc)[:200]}
    return out


def parity_check(eng, plan, step0, bufs, host_in0, base, units, tns, n_streams, n_chan, stream, cce=None):
    """Untimed, after the timed region: reset every stream (= new FilterBank), run the bench's own step once more on
    buffer set 0 and compare the whole batch with the oracle.  Returns (rms error, rms error / signal rms)."""
    import numpy as np
    import torch
    import orc
    torch.cuda.synchronize()
    for s in range(n_streams):
        eng.reset_stream(s)
    d_in, d_out = bufs[0]
    d_out.zero_()
    torch.cuda.synchronize()
    step0()
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    if got.dtype == np.int16:                              # AACG_OUTPUT_I16: compared on the float scale (half a step of rounding on top)
        got = got.astype(np.float32) / np.float32(32768.0)
    ov = np.zeros((n_streams, n_chan + (1 if cce is not None else 0), 1024), np.float32)
    ref = orc.load().decode_batch(units, host_in0, base["meta"] if host_in0.dtype == np.int16 else None, base["n_pcm"], ov, tns=tns, cce=cce)
    if d_out.dtype == torch.int16:                         # int16 PCM saturates (TNS SPEC filters have gain): so does the yardstick
        ref = np.clip(ref, -1.0, 32767.0 / 32768.0)
    d = got.astype(np.float64) - ref
    err, sig = float(np.sqrt(np.mean(d * d))), float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    return err, err / sig if sig > 0 else float("inf"), bool(np.isfinite(got).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--precondition-ms", type=float, default=300.0,
                    help="untimed load before the warm-up steps so that the GPU is at its steady clocks (0 = none)")
    ap.add_argument("--input", choices=["quant", "spec"], default="quant")
    ap.add_argument("--workload", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default="cfg2")
    ap.add_argument("--tns", choices=["reference", "spec"], default="reference",
                    help="spec: AACG_TNS_SPEC engine with TNS side info on every channel-frame as SURVEY config 3 has it (supplementary; "
                         "the reference's TNS is the identity, which is what the headline figure measures)")
    ap.add_argument("--cce", choices=["reference", "spec"], default="reference",
                    help="spec: AACG_CCE_SPEC engine, one independently switched coupling element per frame coupled into 1-4 channels "
                         "(supplementary; the reference parses coupling elements and never applies them)")
    ap.add_argument("--output", choices=["f32", "i16"], default="f32",
                    help="i16: AACG_OUTPUT_I16 engine (supplementary; the reference returns float PCM, which is what the headline measures)")
    ap.add_argument("--repeats", type=int, default=25,
                    help="R: the K-step timed region is run R times back to back, each bracketed by its own HIP events; ms_per_step is the median")
    ap.add_argument("--default-events", action="store_true",
                    help="time with torch.cuda.Event (default HIP events: a system-scope fence per record) instead of timing-only events")
    ap.add_argument("--strict-backend", action="store_true",
                    help="exit non-zero if the backend asked for (--dist-backend, default nccl = RCCL) could not be initialised.  This IS the default "
                         "under torch.distributed.run (the driver's launch line): a gloo fallback there is an error, not a field")
    ap.add_argument("--allow-backend-fallback", action="store_true",
                    help="under torch.distributed.run: fall back to gloo if RCCL cannot be initialised and say so in the line's top-level "
                         "collectives_backend_fallback (tests on boxes without enough GPUs)")
    ap.add_argument("--nbuf", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed oracle comparison after the timed region")
    ap.add_argument("--pipelines", type=int, default=1, choices=[1, 2],
                    help="2: alternate batches of two disjoint stream sets on two HIP streams (supplementary figure; implies --serial)")
    ap.add_argument("--serial", action="store_true",
                    help="launch through aacg_decode_device on one HIP stream, every launch behind the one before it (rounds 1-4) instead "
                         "of aacg_decode_pipelined")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of the harness (barrier + MAX only); gloo for ranks that share a GPU")
    ap.add_argument("--wait-mode", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="measurement only (aacg_debug_set_wait_mode): how the engine's host waits wait — 0 poll without pause (round 5), 1 yield, 2 the shipped "
                         "policy spelled out, 3 blocking events; -1 (default): whatever the library ships")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing on a box with fewer GPUs than ranks: rank r uses device r mod device_count (use with --dist-backend gloo)")
    args = ap.parse_args()

    import aacgpu_shard
    if args.gpus > 1 and not aacgpu_shard.launched_by_torchrun():
        # started without a launcher: run the N ranks as a child process group (nothing here has touched the GPU yet)
        sys.exit(aacgpu_shard.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    # profiling switches of a -DAACG_PROFILE build must never reach a timed run
    if os.environ.pop("AACG_ABLATE", None) is not None:
        print("bench.py: AACG_ABLATE ignored (work-skipping switches are for tools/ only)", file=sys.stderr)
    # the workload generator's diagnostic override of the config-3 window-sequence pattern: allowed (tools/ use it), but never
    # silently — it is recorded in config.workload_override and the line then no longer claims a BASELINE configuration
    seq_override = os.environ.get("AACG_SEQ_PATTERN") or None

    import numpy as np
    import torch
    import aacgpu
    import aacgpu_workload

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local >= n_dev and not args.share_gpu:
        raise SystemExit("rank %d has no GPU of its own (%d visible); --share-gpu is for tests only" % (local, n_dev))
    device = local % n_dev
    torch.cuda.set_device(device)
    dist = None
    if aacgpu_shard.launched_by_torchrun():            # RCCL (or gloo) for barrier / max only
        dist = aacgpu_shard.init_process_group(args.dist_backend, device=torch.device("cuda", device))
        if (args.strict_backend or not args.allow_backend_fallback) and dist.get_backend() != args.dist_backend:
            raise SystemExit("bench.py: --dist-backend %s asked for, %s answered (--strict-backend)" % (args.dist_backend, dist.get_backend()))

    # cfg2 (the metric's configuration) / cfg3: 256 streams x 16 frames; cfg4: 32 streams x 128 frames per GPU
    # (chains of 8 runs: later runs recompute their predecessor's tail); cfg5: 3 CPE + LFE = 7 channels per frame
    mix, n_streams, n_frames, layout, n_chan = workload_shape(args.workload)
    kind = aacgpu.INPUT_QUANT_I16 if args.input == "quant" else aacgpu.INPUT_SPEC_F32
    eng = aacgpu.Engine(kind, max_streams=n_streams * args.pipelines, max_channels=n_chan + (1 if args.cce == "spec" else 0), device=device,
                        cce_mode=aacgpu.CCE_SPEC if args.cce == "spec" else aacgpu.CCE_REFERENCE,
                        tns_mode=aacgpu.TNS_SPEC if args.tns == "spec" else aacgpu.TNS_REFERENCE,
                        output_kind=aacgpu.OUTPUT_I16 if args.output == "i16" else aacgpu.OUTPUT_F32)

    if args.wait_mode >= 0:
        eng.debug_set_wait_mode(args.wait_mode)
    # rank r owns its own streams: independent data per rank, same shape
    base = aacgpu_workload.make_batch(n_streams=n_streams, n_frames=n_frames, mix=mix, layout=layout,
                                      seed=aacgpu_shard.rank_seed(0xAAC00002, rank))
    units, tns, cce = base["units"], None, None
    if args.tns == "spec":                               # a filter on every channel-frame
        units, tns = aacgpu_workload.add_tns_config3(base, seed=0xAAC00003 + rank)
    if args.cce == "spec":                              # BASELINE config 5 names cce.js coupling: one independent CCE per frame
        if args.input != "quant" or tns is not None:
            raise SystemExit("--cce spec: int16 seam, without --tns spec")
        units, cq, cmeta, cce = aacgpu_workload.add_cce(base, points=(2,), seed=0xAAC00005 + rank)
        base = dict(base, q=cq, meta=cmeta)
    plans = []
    for pl in range(args.pipelines):               # pipeline p owns stream slots [p * n_streams, (p + 1) * n_streams)
        up = units.copy()
        up["stream"] += pl * n_streams
        plans.append(eng.plan(up, tns=tns, cce=cce))
    d_meta = torch.from_numpy(base["meta"].view(np.int16)).cuda() if args.input == "quant" else None
    bufs, host_in0 = [], None
    rng = np.random.default_rng(rank)
    for b in range(args.nbuf):
        q = base["q"] if b == 0 else np.roll(base["q"], 131 * b, axis=0) * rng.choice([-1, 1]).astype(np.int16)
        if args.input == "quant":
            h_in = np.ascontiguousarray(q)
        else:
            # filterbank seam: f32 spectra of matching magnitude (IQ * scalefactor of the same data)
            h_in = (np.sign(q) * np.abs(q.astype(np.float32)) ** (4.0 / 3.0) * 2.0 ** 12).astype(np.float32)
        if b == 0:
            host_in0 = h_in
        d_out = torch.empty(base["n_pcm"], dtype=torch.int16 if args.output == "i16" else torch.float32, device="cuda")
        bufs.append((torch.from_numpy(h_in).cuda(), d_out))
    # a dedicated (non-null) stream: kernels, warm-up and the timing events all live on it
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    assert tstream.cuda_stream != 0
    tstreams = [tstream] + [torch.cuda.Stream() for _ in range(args.pipelines - 1)]
    meta_ptr = d_meta.data_ptr() if d_meta is not None else None

    pipelined = not args.serial and args.pipelines == 1

    # The enqueue loop is host work per launch, and with the pipeline's back-pressure the host is part of the loop: the C ABI is
    # called with arguments converted once (ctypes objects per buffer set), not through the convenience wrappers of aacgpu.py.
    import ctypes
    _vp = ctypes.c_void_p
    _ptrs = [(_vp(d_in.data_ptr()), _vp(d_out.data_ptr())) for d_in, d_out in bufs]
    _meta = _vp(meta_ptr) if meta_ptr is not None else None
    _h, _lib, _nbuf = _vp(eng.handle.value if hasattr(eng.handle, "value") else eng.handle), eng.lib, args.nbuf
    _plan_h = [_vp(p.handle.value if hasattr(p.handle, "value") else p.handle) for p in plans]
    _piped, _piped_timed, _serial = _lib.aacg_decode_pipelined, _lib.aacg_decode_pipelined_timed, _lib.aacg_decode_device
    _streams = [_vp(t.cuda_stream) for t in tstreams]

    def step(i, mark=None):
        a, o = _ptrs[i % _nbuf]
        if pipelined:                                  # the engine's internal streams in turn: launch i may overlap launch i - 1
            rc = _piped(_h, _plan_h[0], a, _meta, o) if mark is None else _piped_timed(_h, _plan_h[0], a, _meta, o, mark.h)
        else:
            pl = i % args.pipelines
            rc = _serial(_h, _plan_h[pl], a, _meta, o, _streams[pl])
        if rc:
            eng._check(rc)

    def join():                                        # the timing stream behind everything launched so far
        if pipelined:
            eng.pipeline_join(tstream.cuda_stream)
        for extra in tstreams[1:]:
            tstream.wait_stream(extra)

    # The GPU reaches its steady clocks only after tens of milliseconds of load: a 4096-frame step takes ~13 us,
    # so a few hundred warm-up steps are over before the clocks have ramped (measured: 15.9 us per step after 40
    # warm-up steps, 14.2 us after 4000).  Untimed preconditioning, reported in the JSON line; then the W warm-up
    # steps of the contract; the timed region is exactly K steps.
    # R repeats of the K-step region.  With overlapped launches a repeat's end is the completion of launches that run side by
    # side with the next repeat's first ones, so short regions scatter by a launch or so either way around the same mean
    # (K = 20: 9.2-12.8 us per step over 25 repeats, mean 11.43 both times, medians 11.22 and 11.89 in two runs): short
    # regions are repeated more often, until about 2000 launches are timed — the median settles, the run stays short.
    R = max(1, args.repeats, min(200, -(-2000 // max(1, args.steps))) if args.repeats == 25 else args.repeats)
    # Timing marks: HIP events that only measure time (hipEventDisableSystemFence, aacg_timer_*).  A default event's record is
    # a system-scope fence — cache write-back and invalidation — and the launch behind it starts on cold caches: at the
    # driver's K = 20 that is one fence per 240 us of launches (12.45 against 12.13 us per step at K = 2000 on one box).
    # --default-events measures with torch.cuda.Event instead.
    class _Mark:
        def __init__(self):
            self.ev = torch.cuda.Event(enable_timing=True) if args.default_events else aacgpu.TimerMark()
        def record(self):
            if args.default_events: self.ev.record(tstream)
            else: self.ev.record(tstream.cuda_stream)
        def elapsed_time(self, later):
            return self.ev.elapsed_time(later.ev) if args.default_events else self.ev.elapsed_ms(later.ev)
    evs = [_Mark() for _ in range(R + 1)]
    issued = [0.0]
    # Pipelined launches: a repeat ends when its last launch AND the ones that run beside it on the other streams are
    # complete.  Their marks are bound to the dispatches themselves (aacg_decode_pipelined_timed: the time stamp is the
    # dispatch's end, no marker packet enters a queue, nothing is joined inside the timed region); the opening mark is an
    # ordinary one on the timing stream, joined behind the warm-up steps.
    bound = pipelined and not args.default_events
    # Regions shorter than 100 launches: the marks go behind every G-th repeat (G x K >= 100 launches), not behind every one.  The median
    # was taken over such runs of consecutive repeats anyway (region_stats says why: a 20-launch repeat's end scatters by a launch
    # either way); what round 6 found is that the marks themselves are not free when they are dense — three timing events bound to
    # every twenty launches cost the overlapped route 1.1-1.5 us per launch on a box whose default line reads 11.0 (12.2-12.6 at the
    # driver's K = 20; with K = 100, three marks per hundred launches: 10.9-11.0; `--default-events`: 11.3): the same K-step region,
    # the same R x K launches back to back, fewer time stamps taken inside them.
    group = -(-100 // args.steps) if (bound and args.steps < 100) else 1
    if group > 1:
        R = -(-R // group) * group
    n_marked = R // group
    tails = [[aacgpu.TimerMark() for _ in range(min(aacgpu.PIPE_STREAMS, args.steps))] for _ in range(n_marked)] if bound else None
    flat_marks = [m for t in tails for m in t] if bound else []

    n_pre = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
        for _ in range(256):
            # (every timing mark is bound once here, untimed: a mark's FIRST binding to a dispatch costs the host more than its
            # later ones — tools: 11.13 us per launch with three fresh marks per twenty launches, 10.97 with the same marks again)
            step(n_pre, flat_marks[n_pre] if n_pre < len(flat_marks) else None)
            n_pre += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(n_pre + i)
    def timed_steps():
        t0 = time.perf_counter()
        join()                                           # the opening mark: the warm-up steps are complete
        evs[0].record()
        K = args.steps
        for r in range(R):
            marks = tails[r // group] if (bound and r % group == group - 1) else None      # the repeat that closes a timed run
            base_i = n_pre + args.warmup + r * K
            for i in range(K):
                k = K - 1 - i                            # 0 for the repeat's last launch, 1 for the one before it, ...
                step(base_i + i, marks[k] if marks is not None and k < len(marks) else None)
            if not bound:
                join()                                   # a mark's time stamp is the completion of every launch before it
                evs[r + 1].record()
        issued[0] = time.perf_counter() - t0             # host time to enqueue the R x K launches

    dev = torch.device("cuda", device)
    _, wall = aacgpu_shard.timed(dist, torch.cuda.synchronize, timed_steps, dev)      # barrier + synchronize on both sides, MAX over ranks
    if bound:
        ends = [0.0] + [max(evs[0].ev.elapsed_ms(m) for m in tails[g]) for g in range(n_marked)]      # ms since the opening mark
        mine_ms = [ends[g + 1] - ends[g] for g in range(n_marked)]                   # runs of `group` consecutive repeats of K steps
    else:
        mine_ms = [evs[r].elapsed_time(evs[r + 1]) for r in range(R)]                 # this rank's R repeats of K steps, on the launch stream
    region_ms = aacgpu_shard.reduce_max_list(dist, mine_ms, dev)                      # MAX over ranks, timed run by timed run
    stats = region_stats(region_ms, args.steps * group, 1)
    if group > 1:
        stats["median_over"] = "%d timed runs of %d consecutive repeats of the %d-step region (%d launches each): marks behind every %d-th repeat" % (n_marked, group, args.steps, group * args.steps, group)
        stats["repeats_of_the_k_step_region"] = R
    kernel_ms = region_stats(mine_ms, args.steps * group, 1)["ms_per_step_median"]    # rank 0's own launches, for its roofline

    # Same process, same box, same clocks, right behind the timed region (untimed itself): what this box's memory system
    # gives a float4 copy launch of the step's byte volume (half read, half written; buffers rotated past the Infinity
    # Cache like the step's), and a 1 GiB copy for the streaming rate.  Boxes of the pool differ by several per cent
    # (DESIGN.md 5); `frac_of_copy` = achieved / copy_ceiling is the figure that does not move with them.
    def copy_rate(n_bytes, n_sets, reps, repeats):
        n_bytes = int(n_bytes) // 16 * 16
        src = [torch.empty(n_bytes, dtype=torch.uint8, device="cuda").random_(0, 255) for _ in range(n_sets)]
        dst = [torch.empty(n_bytes, dtype=torch.uint8, device="cuda") for _ in range(n_sets)]
        ce = [_Mark() for _ in range(repeats + 1)]
        for i in range(max(2, reps // 10)):
            aacgpu.calib_copy(dst[i % n_sets].data_ptr(), src[i % n_sets].data_ptr(), n_bytes, tstream.cuda_stream)
        ce[0].record()
        for r in range(repeats):
            for i in range(reps):
                aacgpu.calib_copy(dst[i % n_sets].data_ptr(), src[i % n_sets].data_ptr(), n_bytes, tstream.cuda_stream)
            ce[r + 1].record()
        torch.cuda.synchronize()
        st = region_stats([ce[r].elapsed_time(ce[r + 1]) for r in range(repeats)], reps)
        return 2.0 * n_bytes / (st["ms_per_step_median"] * 1e-3) / 1e9, st

    abytes_step = algorithmic_bytes_per_channel_frame(args.input, n_frames, args.output) * n_streams * n_frames * n_chan
    copy_reps = max(20, min(args.steps, 400))
    copy_gbs, copy_stats = copy_rate(abytes_step / 2, max(2, args.nbuf), copy_reps, R)
    copy_ms = copy_stats["ms_per_step_median"]
    copy_large_gbs, _ = copy_rate(1 << 29, 2, 20, 1)

    # untimed: the last output is finite and non-trivial; then the oracle comparison on the bench's own batch
    out = bufs[(n_pre + args.warmup + R * args.steps - 1) % args.nbuf][1]
    ok = bool(torch.isfinite(out.float()).all().item()) and float(out.float().abs().max().item()) > 0
    parity = None
    if not args.no_parity:
        err, rel, finite = parity_check(eng, plans[0], lambda: step(0), bufs, host_in0, base, units, tns, n_streams, n_chan, tstream, cce)
        gate_rel = 5e-6 if tns is None else 1e-5         # AACG_TNS_SPEC: DESIGN.md §3a
        gate_rms = 1e-5
        if args.output == "i16":                          # rounding to 1 / 32768: 8.8e-6 rms of quantisation noise by itself
            gate_rms, gate_rel = 2e-5, 2e-4
        parity = {"rms": err, "rel": rel, "gate_rms": gate_rms, "gate_rel": gate_rel, "frames": n_streams * n_frames,
                  "against": "oracle/aac_oracle.c on the whole batch after aacg_reset_stream on every stream"}
        ok = ok and finite and err <= gate_rms and rel <= gate_rel

    frames_per_step = n_streams * n_frames
    value = whole_job_value(world, frames_per_step, stats["ms_per_step_median"])
    abytes = algorithmic_bytes_per_channel_frame(args.input, n_frames, args.output) * frames_per_step * n_chan
    if cce is not None:
        abytes += (2048 + 240) * frames_per_step             # the coupling element's own spectrum and band words in, nothing extra out
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    kernel_names = eng.plan_kernels(plans[0], pipelined=pipelined)
    traffic, traffic_src = measured_traffic(kernel_names) if args.workload == "cfg2" else (None, None)
    unit = "stereo frames/s" if n_chan == 2 else "7-channel frames/s"
    line = {
        "metric": "AAC-LC 48 kHz stereo frames/sec per node + achieved HBM GB/s vs roofline",
        "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": stats["ms_per_step_median"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timing": dict(stats, events=("torch.cuda.Event (default HIP events: a system-scope fence per record)" if args.default_events else
                                      "hipEventDisableSystemFence (timing-only HIP events, aacg_timer_*)"),
                       method=("the K timed steps are run R times back to back" + (" and timed in runs of %d consecutive repeats (%d launches)" % (group, group * args.steps) if group > 1 else "") +
                               "; a timed run ends when its last launches (one per pipeline stream: they "
                               "run side by side) are all complete — HIP events bound to those dispatches' completion "
                               "(hipExtLaunchKernel stopEvent), the opening event on the timing stream behind the warm-up steps" if bound else
                               "the K timed steps are run R times back to back, each repeat between its own HIP events on the launch stream") +
                              " (MAX over ranks per repeat); value and ms_per_step: the MEDIAN repeat; wall_*: host clock between "
                              "the barriers over all R x K steps (adds first-launch latency and the wake-up after the last step)"),
        "wall_ms_per_step": wall / (R * args.steps) * 1e3, "value_wall": world * frames_per_step * R * args.steps / wall,
        "config": {"workload": {"cfg2": "BASELINE config 2: batch of 4096 stereo LC frames (256 streams x 16 frames, ONLY_LONG_SEQUENCE, KBD)",
                                "cfg3": "BASELINE config 3: 4096 stereo frames, window-sequence mix [0,0,1,2,2,3,0,0], TNS identity",
                                "cfg4": "BASELINE config 4 shape per GPU: 32 streams x 128 frames, config-3 mix",
                                "cfg5": "BASELINE config 5 shape per GPU: 4096 frames of 3 CPE + LFE (7 channels), config-3 mix"}[args.workload],
                   "workload_override": ("AACG_SEQ_PATTERN=%s: NOT the BASELINE window-sequence mix" % seq_override) if (seq_override and mix) else None,
                   "input": "int16 quantised spectra + band side info (process(elements) seam)" if args.input == "quant"
                   else "f32 spectra (FilterBank.process seam)",
                   "output": "float32 PCM as the reference returns it" if args.output == "f32" else "int16 PCM (AACG_OUTPUT_I16)",
                   "streams_per_gpu": n_streams, "frames_per_stream_per_step": n_frames, "buffers_rotated": args.nbuf,
                   "realtime_multiple": value / 46.875, "sharding": "streams over ranks, no data-path collective", "pipelines": args.pipelines,
                   "launches": ("aacg_decode_pipelined: ONE plan, the same %d streams continued launch after launch on %d of the engine's internal HIP "
                                "streams taken in turn; consecutive launches overlap, their chains meet in rendezvous cells (nobody waits); %d of the %d "
                                "launches of this process continued the launch before them; the streams were %s" % (n_streams, eng.pipeline_streams_used(), eng.pipeline_chained(), n_pre + args.warmup + R * args.steps + (0 if args.no_parity else 1),
                                   "seen to run side by side when the pipeline was set up" if eng.pipeline_concurrent() else "NOT seen to run side by side (one hardware queue): the launches serialise"))
                               if pipelined else "aacg_decode_device: every launch behind the one before it on one HIP stream",
                   "preconditioning": "%d untimed steps (%.0f ms of load) before the warm-up steps: steady GPU clocks" % (n_pre, args.precondition_ms),
                   "tns": "identity, as the reference executes it" if tns is None
                          else "AACG_TNS_SPEC, every channel-frame: long one filter of order 12 over 20 bands, short one of order 7 per window",
                   "coupling": "none applied, as the reference executes it" if cce is None
                               else "AACG_CCE_SPEC: one independently switched coupling element per frame into 1-4 channels",
                   "collectives": "none on the data path; %s" % (("%s barrier + 8-byte MAX around the timed region, world size %d as the backend reports it"
                                                                   % (dist.get_backend(), dist.get_world_size())) if dist is not None else "single process, no process group")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_run": False,
                     "copy_ceiling_GBs": copy_gbs, "frac_of_copy": achieved / copy_gbs, "copy_ms": copy_ms,
                     "copy_timing": copy_stats,
                     "copy_ceiling_note": "aacg_calib_copy: float4 copy of the step's algorithmic byte volume (half read, half written) with the run "
                                          "kernel's launch shape, %d x %d launches right behind the timed region on the same stream (median repeat); 1 GiB copy: %.0f GB/s" % (R, copy_reps, copy_large_gbs),
                     "copy_ceiling_large_GBs": copy_large_gbs,
                     "kernel": kernel_names,
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": abytes,
                     "kernel_ms_note": ("per-launch time of the timed region (K launches between two marks) — with overlapped launches the "
                                        "START-TO-START interval of consecutive dispatches, which is what a rocprofv3 kernel trace shows as the "
                                        "spacing of the rows (profiles/rNN_*_intervals.txt); one dispatch's own begin-to-end duration is longer, "
                                        "two of them are in flight at a time") if pipelined else "per-launch time of the timed region = the dispatch duration a kernel trace shows",
                     "host_enqueue_us_per_step": issued[0] / (R * args.steps) * 1e6},
        "output_ok": ok, "parity_rms": parity["rms"] if parity else None, "parity": parity,
    }
    line.update(backend_fields(args.dist_backend, dist))
    if rank == 0 and not args.no_cpu_baseline:
        # N = 1: all host cores, one core, and the JavaScript port; N > 1: one core (the field is never null on a line the driver records)
        line["cpu_baseline"] = cpu_baseline(args.input, mix, layout, n_chan) if world == 1 else cpu_baseline(args.input, mix, layout, n_chan, budget_s=5.0, all_cores=False)
        line["cpu_baseline"]["unit"] = line["cpu_baseline"]["single_core"]["unit"] = unit
        if "js_port" in line["cpu_baseline"] and "unit" in line["cpu_baseline"]["js_port"]:
            line["cpu_baseline"]["js_port"]["unit"] = unit
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    for pl in plans:
        pl.destroy()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: output check failed (finite=%s, parity=%s)" % (ok, parity))


if __name__ == "__main__":
    main()
