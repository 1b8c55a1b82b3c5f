#!/usr/bin/env python3
"""Install one `tools/collect_round.sh <tag>` collection (gpurun_out/<tag>/) as the round's committed evidence:
    python tools/install_round.py r05y r05
copies bench_*.json -> profiles/<round>_bench_*.json, the rocprofv3 summaries / kernel stats / dispatch spacings ->
profiles/<round>_*_{summary.txt,kernel_stats.csv,intervals.txt}, the plugin-surface rate, the microbenchmarks and the timelines,
builds profiles/<round>_traffic.json from the PMC passes of the same box (FETCH_SIZE x 2 + WRITE_SIZE, KiB: MI355X_MICROARCH.md,
HBM section) keyed by the kernel that was PROFILED, writes those bytes into the lines whose kernel it is, and regenerates
DESIGN.md section 5's table and README.md's headline sentence from the installed lines.  Narrative numbers elsewhere in the docs
stay the author's to check."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = 1024


def line_of(path):
    with open(path) as f:
        return json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])


def pmc(summary, kernel):
    """(FETCH_SIZE, WRITE_SIZE, TCP_TCC_WRITE_REQ_sum) per launch of `kernel`: its "== PMC per dispatch" block of a tools/prof.sh summary."""
    blocks = open(summary).read().split("== PMC per dispatch (mean over dispatches): ")
    out = {"FETCH_SIZE": None, "WRITE_SIZE": None, "TCP_TCC_WRITE_REQ_sum": None}
    for blk in blocks[1:]:
        if blk.splitlines()[0].strip() != kernel:
            continue
        for name in out:
            m = re.search(r"^\s*" + name + r"\s+([0-9.]+)", blk, re.M)
            out[name] = float(m.group(1)) if m else None
    return out


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    prof = os.path.join(ROOT, "profiles")
    profiled = {"quant": ("prof_quant", "aacg_imdct_run_quant_rv", "cfg2"), "quant_serial": ("prof_quant_serial", "aacg_imdct_run_quant", "cfg2"),
                "spec": ("prof_spec", "aacg_imdct_run_f32_rv", "cfg2"), "cfg5_quant": ("prof_cfg5", "aacg_imdct_run_quant_rv_nt", "cfg5"),
                "cfg3_tns_quant_ex": ("prof_cfg3_tns", "aacg_imdct_run_quant_ex_rv", "cfg3")}
    for p_ in sorted(set(v[0] for v in profiled.values())):
        for suffix in ("summary.txt", "kernel_stats.csv", "intervals.txt"):
            f = os.path.join(src, "%s_%s" % (p_, suffix))
            if os.path.exists(f):
                shutil.copy(f, os.path.join(prof, "%s_%s_%s" % (rnd, p_[5:], suffix)))
        # overlapped launches: a trace row's duration includes its wait for CUs and the launches running beside it; what a launch
        # COSTS is the spacing of the rows — say so at the top of the summary, with the figures of the same trace
        summ, iv = os.path.join(prof, "%s_%s_summary.txt" % (rnd, p_[5:])), os.path.join(src, "%s_intervals.txt" % p_)
        if os.path.exists(summ) and os.path.exists(iv):
            lines = [l for l in open(iv).read().splitlines() if l.startswith(("kernel:", "dispatch duration", "overlap with", "time per launch", "NOTE in flight", "driver:"))]
            over = any(l.startswith("overlap with") and "median 0.00" not in l for l in lines)
            if over:
                note = ["NOTE  launches of this route OVERLAP (aacg_decode_pipelined, DESIGN.md 3d): avg_ns below is the length of a trace row (begin -> end,",
                        "NOTE  including the row's wait for CUs), not what a launch costs; the same trace's row spacing (tools/kernel_intervals.py, %s_%s_intervals.txt):" % (rnd, p_[5:])]
                body = open(summ).read()
                if not body.startswith("NOTE"):
                    open(summ, "w").write("\n".join(note + ["NOTE  " + l for l in lines]) + "\n" + body)
    for f, dst in (("readchunk_256streams.json", "readchunk_256streams.json"), ("readchunk_256streams_surround.json", "readchunk_256streams_surround.json"), ("micro.txt", "micro.txt"),
                   ("timeline.txt", "timeline.txt"), ("pipe_drive.jsonl", "pipe_drive.jsonl"), ("resident.jsonl", "resident_drive.jsonl"), ("wait_modes.txt", "wait_modes.txt"), ("resident_budget.txt", "resident_budget.txt"), ("dequant_floor.txt", "dequant_floor.txt"),
                   ("prof_quant_bench_intervals.txt", "quant_bench_intervals.txt"), ("prof_quant_bench_kernel_stats.csv", "quant_bench_kernel_stats.csv")):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(prof, "%s_%s" % (rnd, dst)))
    T = {"note": "HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB; separate passes, tools/prof.sh, profiles/%s_*_summary.txt). gfx950 correction per "
                 "MI355X_MICROARCH.md section HBM: FETCH_SIZE reports 1/2 of a wide (16 B/lane) coalesced read -> x2; WRITE_SIZE as is. Both checked in round 1 on a launch with "
                 "known bytes (profiles/r01_calib_memonly_summary.txt: ratios 0.513 and 1.001). Keyed by the kernel that was profiled: bench.py quotes a record only for that kernel." % rnd}
    for key, (p_, kernel, workload) in profiled.items():
        c = pmc(os.path.join(prof, "%s_%s_summary.txt" % (rnd, p_[5:])), kernel)
        if c["FETCH_SIZE"] is None or c["WRITE_SIZE"] is None:
            print("no PMC values for", key, kernel, file=sys.stderr)
            continue
        T[key] = {"kernel": kernel, "workload": workload, "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                  "traffic_bytes": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * K}
        if c["TCP_TCC_WRITE_REQ_sum"]:
            T[key]["TCP_TCC_WRITE_REQ"] = int(round(c["TCP_TCC_WRITE_REQ_sum"]))
    tpath = os.path.join(prof, rnd + "_traffic.json")
    json.dump(T, open(tpath, "w"), indent=1)
    by_kernel = {(v["kernel"], v["workload"]): v for v in T.values() if isinstance(v, dict)}
    for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        lines = open(f).read().splitlines()
        for i, l in enumerate(lines):
            if not l.startswith("{"):
                continue
            d = json.loads(l)
            r = d["roofline"]
            wl = "cfg2" if "config 2" in d["config"]["workload"] else ("cfg5" if "config 5" in d["config"]["workload"] else ("cfg3" if "config 3" in d["config"]["workload"] else "cfg4"))
            rec = by_kernel.get((r["kernel"], wl)) if (d["config"].get("tns", "").startswith("identity") or "_ex" in r["kernel"]) else None
            r["traffic"], r["traffic_source"] = (float(rec["traffic_bytes"]), "profiles/%s_traffic.json" % rnd) if rec else (None, None)
            lines[i] = json.dumps(d)
        open(os.path.join(prof, "%s_%s" % (rnd, os.path.basename(f))), "w").write("\n".join(lines) + "\n")

    def L(n):
        return line_of(os.path.join(prof, "%s_bench_%s.json" % (rnd, n)))
    mb = lambda k: T[k]["traffic_bytes"] / 1e6
    x_ = lambda k, n: "%.1f MB (%.2f× algorithmic)" % (mb(k), T[k]["traffic_bytes"] / L(n)["roofline"]["algorithmic_bytes_per_launch"]) if k in T else ""
    rows = [("int16 in → f32 PCM, config 2, launches overlapped (`aacg_decode_pipelined`) — **headline**", "quant", x_("quant", "quant")),
            ("the same at the driver's `--steps 20 --warmup 5`", "quant_20steps", ""),
            ("the same, launch behind launch on one stream (`--serial`: rounds 1-4's method, the plain kernel)", "quant_serial", x_("quant_serial", "quant_serial")),
            ("`--serial` at `--steps 20 --warmup 5`", "quant_serial_20steps", ""),
            ("f32 in (filterbank seam), config 2, overlapped", "spec", x_("spec", "spec")),
            ("f32 in, `--serial`", "spec_serial", ""),
            ("int16 in, config 3 (all window sequences mixed), overlapped", "cfg3", ""),
            ("config 3, `--serial`", "cfg3_serial", ""),
            ("config 4 shape (32 streams × 128 frames per GPU: 8 runs per chain, rendezvous between runs AND between launches), overlapped", "cfg4", ""),
            ("config 4 shape, `--serial` (rendezvous between runs only)", "cfg4_serial", ""),
            ("config 5 shape (7 channels per frame), int16 in, overlapped", "cfg5", ("%s; %.1f M L2 write requests" % (x_("cfg5_quant", "cfg5"), T["cfg5_quant"].get("TCP_TCC_WRITE_REQ", 0) / 1e6)) if "cfg5_quant" in T else ""),
            ("config 5 shape, `--serial`", "cfg5_serial", ""),
            ("config 5 shape, f32 in, overlapped", "cfg5_spec", ""),
            ("int16 in → **int16 PCM**, config 2, overlapped (`aacg_imdct_run_quant_rv_i16`)", "quant_i16out", ""),
            ("int16 PCM, `--serial`", "quant_i16out_serial", ""),
            ("config 3 + `AACG_TNS_SPEC`, a filter on every channel-frame, overlapped (`aacg_imdct_run_quant_ex_rv`)", "cfg3_tns_spec_quant", x_("cfg3_tns_quant_ex", "cfg3_tns_spec_quant")),
            ("the same, f32 seam", "cfg3_tns_spec_f32", ""),
            ("config 3 + `AACG_TNS_SPEC`, `--serial`", "cfg3_tns_spec_quant_serial", ""),
            ("config 5 + `AACG_CCE_SPEC`, one independent CCE per frame (two launches, not overlapped)", "cfg5_cce_spec", ""),
            ("two disjoint stream sets on two HIP streams, plain kernel (`--pipelines 2`, supplementary: what overlap is worth without a rendezvous)", "quant_pipelines2", ""),
            ("driver's `torch.distributed.run` line, one rank over RCCL", "quant_torchrun_rccl_1rank", "")]
    out = ["| path | µs / launch: median (min – max of 25 repeats) | frames/s | achieved | of 8 TB/s | of the same-run copy | PMC traffic / launch | parity vs oracle (rms) |", "|---|---|---|---|---|---|---|---|"]
    for name, key, tr in rows:
        try:
            x = L(key)
        except (OSError, IndexError):
            continue
        tm = x["timing"]; r = x["roofline"]
        out.append("| %s | **%.2f** (%.2f – %.2f) | %.1f M | %.2f TB/s | %.1f %% | %.2f of %.2f TB/s | %s | %.1e |" % (
            name, tm["ms_per_step_median"] * 1e3, tm["ms_per_step_min"] * 1e3, tm["ms_per_step_max"] * 1e3, x["value"] / 1e6, r["achieved"] / 1e3,
            100 * r["frac"], r["frac_of_copy"], r["copy_ceiling_GBs"] / 1e3, tr, x["parity_rms"]))
    q = L("quant"); cb = q["cpu_baseline"]
    tab = "\n".join(out) + "\n\nCPU baselines of the headline's run (`cpu_baseline`, `kind: port`): %.0f k stereo frames/s on all %d host threads, %.1f k on one, %.1f k for the JavaScript port under Node." % (
        cb["value"] / 1e3, cb["cores"], cb["single_core"]["value"] / 1e3, cb.get("js_port", {}).get("value", 0) / 1e3)
    p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
    a = s.index("| path | µs / launch: median"); b = s.index("PMC traffic = FETCH_SIZE × 2")
    open(p, "w").write(s[:a] + tab + "\n\n" + s[b:])
    p = os.path.join(ROOT, "README.md"); s = open(p).read()
    head = "%.1f M stereo frames/s (%.1f M× real time), %.2f µs per 4096-frame launch, %.2f TB/s of algorithmic traffic = %.1f %% of the 8 TB/s HBM peak and %.2f of a copy kernel of the same byte volume timed in the same run" % (
        q["value"] / 1e6, q["config"]["realtime_multiple"] / 1e6, q["ms_per_step"] * 1e3, q["roofline"]["achieved"] / 1e3, 100 * q["roofline"]["frac"], q["roofline"]["frac_of_copy"])
    s = re.sub(r"Measured on one MI355X \(round \d+, `profiles/r\d+_\*`, DESIGN.md §5\): .*? on the\nint16 seam",
               "Measured on one MI355X (round %d, `profiles/%s_*`, DESIGN.md §5): " % (int(rnd[1:]), rnd) + head + " on the\nint16 seam", s, flags=re.S)
    open(p, "w").write(s)
    print(head)
    for name, key, _ in rows:
        try:
            x = L(key)
        except (OSError, IndexError):
            continue
        print("%-28s %-34s %.3f us  frac %.3f  of copy %.3f" % (key, x["roofline"]["kernel"][:34], x["roofline"]["kernel_ms"] * 1e3, x["roofline"]["frac"], x["roofline"]["frac_of_copy"]))


if __name__ == "__main__":
    main()
