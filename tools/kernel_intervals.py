#!/usr/bin/env python3
"""Start-to-start spacing and overlap of consecutive dispatches from a rocprofv3 --kernel-trace CSV.

With aacg_decode_pipelined two launches of the run kernel are in flight at a time: one dispatch's own begin-to-end duration
(what `--stats` averages) is then LONGER than the time a launch costs — the figure that matters is the spacing of the rows.
usage: kernel_intervals.py <dir or kernel_trace.csv> [kernel substring] [--regions]

--regions: the trace is cut where no dispatch of the kernel was running for more than 50 us (a drained pipeline between two timed
regions of tools/micro/pipe_drive); regions of fewer than 1000 dispatches (preconditioning bursts) are dropped, a twentieth is
trimmed off both ends of every region, and the figures are taken over what is left."""
import csv
import glob
import os
import sys


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    by_region = "--regions" in sys.argv
    path = argv[0]
    want = argv[1] if len(argv) > 1 else "aacg_imdct_run"
    files = [path] if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    if len(rows) < 10:
        print("no dispatches of", want, "in", files)
        return
    if by_region:
        return regions(rows)
    # steady part: skip the first and last tenth
    lo, hi = len(rows) // 10, len(rows) - len(rows) // 10
    part = rows[lo:hi]
    dur = sorted(e - s for s, e, _, _ in part)
    s2s = sorted(part[i + 1][0] - part[i][0] for i in range(len(part) - 1))
    e2e = sorted(part[i + 1][1] - part[i][1] for i in range(len(part) - 1))
    ov = sorted(max(0, part[i][1] - part[i + 1][0]) for i in range(len(part) - 1))
    gap = sorted(max(0, part[i + 1][0] - part[i][1]) for i in range(len(part) - 1))
    med = lambda a: a[len(a) // 2]
    span = (part[-1][1] - part[0][0]) / (len(part) - 1 + (part[-1][1] - part[-1][0]) / max(1, med(s2s)))
    print("kernel: %s   dispatches: %d (steady part %d..%d)   queues: %s" % (part[0][2], len(rows), lo, hi, sorted(set(q for *_, q in part))))
    print("dispatch duration (begin -> end)  median %.2f us  min %.2f  max %.2f  mean %.2f" % (med(dur) / 1e3, dur[0] / 1e3, dur[-1] / 1e3, sum(dur) / len(dur) / 1e3))
    print("start -> next start               median %.2f us  mean %.2f" % (med(s2s) / 1e3, sum(s2s) / len(s2s) / 1e3))
    print("end   -> next end                 median %.2f us  mean %.2f" % (med(e2e) / 1e3, sum(e2e) / len(e2e) / 1e3))
    print("overlap with the next dispatch    median %.2f us  mean %.2f   (0 = serialised)" % (med(ov) / 1e3, sum(ov) / len(ov) / 1e3))
    print("gap to the next dispatch          median %.2f us  mean %.2f" % (med(gap) / 1e3, sum(gap) / len(gap) / 1e3))
    print("time per launch over the steady part: %.2f us  [(last end - first start) / launches]" % ((part[-1][1] - part[0][0]) / len(part) / 1e3))


def regions(rows):
    cuts, run, busy_until = [], [rows[0]], rows[0][1]
    for r in rows[1:]:
        if r[0] - busy_until > 50000:
            cuts.append(run)
            run = []
        run.append(r)
        busy_until = max(busy_until, r[1])
    cuts.append(run)
    keep = [c[len(c) // 20: len(c) - len(c) // 20] for c in cuts if len(c) >= 1000]
    if not keep:
        print("no region of 1000 dispatches or more")
        return
    med = lambda a: sorted(a)[len(a) // 2]
    dur = [e - s for c in keep for s, e, _, _ in c]
    s2s = [c[i + 1][0] - c[i][0] for c in keep for i in range(len(c) - 1)]
    e2e = [c[i + 1][1] - c[i][1] for c in keep for i in range(len(c) - 1)]
    per_region = [(c[-1][1] - c[0][1]) / (len(c) - 1) for c in keep]          # end of the first kept dispatch to end of the last
    n = sum(len(c) for c in keep)
    per_launch = sum(c[-1][1] - c[0][1] for c in keep) / sum(len(c) - 1 for c in keep)
    # dispatches in flight: at every start, how many earlier dispatches have not ended yet (+ itself)
    inflight = []
    for c in keep:
        ends = []
        for s, e, _, _ in c:
            ends = [x for x in ends if x > s]
            ends.append(e)
            inflight.append(len(ends))
    print("kernel: %s   dispatches: %d in the trace, %d regions of >= 1000 (kept %d, a twentieth trimmed off both ends of each)   queues: %s"
          % (keep[0][0][2], len(rows), len(keep), n, sorted(set(q for c in keep for *_, q in c))))
    print("dispatch duration (begin -> end)  median %.2f us  min %.2f  max %.2f  mean %.2f" % (med(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3, sum(dur) / len(dur) / 1e3))
    print("start -> next start               median %.2f us  mean %.2f" % (med(s2s) / 1e3, sum(s2s) / len(s2s) / 1e3))
    print("end   -> next end                 median %.2f us  mean %.2f" % (med(e2e) / 1e3, sum(e2e) / len(e2e) / 1e3))
    ov = [max(0, c[i][1] - c[i + 1][0]) for c in keep for i in range(len(c) - 1)]
    print("overlap with the next dispatch    median %.2f us  mean %.2f   (0 = serialised)" % (med(ov) / 1e3, sum(ov) / len(ov) / 1e3))
    print("time per launch, region by region: %s us" % " ".join("%.2f" % (t / 1e3) for t in per_region))
    print("time per launch over the steady part: %.2f us  [(last end - first end) / (launches - 1), all kept regions]" % (per_launch / 1e3))
    print("NOTE in flight: %.2f dispatches on average (mean duration / time per launch = %.2f / %.2f), %.2f counted at the dispatches' starts (median %d);"
          % (sum(dur) / len(dur) / per_launch, sum(dur) / len(dur) / 1e3, per_launch / 1e3, sum(inflight) / len(inflight), med(inflight)))
    print("NOTE a --stats row (the dispatch duration) is therefore that many times what a launch costs: bytes / (mean duration / in flight) = bytes / time per launch")


if __name__ == "__main__":
    main()
