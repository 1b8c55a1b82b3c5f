#!/usr/bin/env python3
"""Start-to-start spacing and overlap of consecutive dispatches from a rocprofv3 --kernel-trace CSV.

With aacg_decode_pipelined two launches of the run kernel are in flight at a time: one dispatch's own begin-to-end duration
(what `--stats` averages) is then LONGER than the time a launch costs — the figure that matters is the spacing of the rows.
usage: kernel_intervals.py <dir or kernel_trace.csv> [kernel substring]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "aacg_imdct_run"
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


if __name__ == "__main__":
    main()
