#!/usr/bin/env python3
"""Where a batch's time goes on the resident route (aacg_pipeline_submit / collect), from ONE rocprofv3 run with
--kernel-trace --memory-copy-trace over tools/micro/resident_drive:

    python tools/resident_budget.py <kernel_trace.csv> <memory_copy_trace.csv> [label]

Per step of a batch (bytes up, the parser's four kernels, the unit records' refresh, the transform, PCM down): how long one
instance takes (median, begin -> end of the trace row) — and, for the whole run: the time between two PCM copies' ends (what a
batch COSTS), how busy the link's downward direction was, how many batches' steps overlapped.  Steady state only: the first
and last tenth of the PCM copies are left out."""
import csv
import statistics
import sys


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def med(v):
    return statistics.median(v) if v else float("nan")


def main():
    kern, cop = rows(sys.argv[1]), rows(sys.argv[2])
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    # the batches' PCM copies: the big device-to-host copies (a batch's results travel by kernel)
    d2h = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cop if r["Direction"].endswith("DEVICE_TO_HOST")), key=lambda x: x[1])
    if len(d2h) < 20:
        sys.exit("too few PCM copies in the trace")
    long_ = med([e - s for s, e in d2h])
    d2h = [x for x in d2h if x[1] - x[0] > 0.3 * long_]
    n = len(d2h)
    cut = max(2, n // 10)
    keep = d2h[cut:n - cut]
    t0, t1 = keep[0][0], keep[-1][1]
    period = (keep[-1][1] - keep[0][1]) / (len(keep) - 1)
    busy = sum(e - s for s, e in keep) / (t1 - t0)
    print("resident route, %s: %d batches in the trace, %d kept (steady state)" % (label, n, len(keep)))
    print("  a batch costs (end of a PCM copy -> end of the next)   %8.1f us" % (period / 1e3))
    print("  PCM copy down, one per batch (begin -> end)            %8.1f us median   link's downward direction busy %.0f %% of the time" % (med([e - s for s, e in keep]) / 1e3, 100 * busy))
    h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cop if r["Direction"].endswith("HOST_TO_DEVICE") and t0 <= int(r["Start_Timestamp"]) <= t1]
    if len(h2d) > len(keep) // 2:
        print("  runtime copies UP in the window: %d, %.1f us median — the bytes go through the SDMA engines in this build" % (len(h2d), med([e - s for s, e in h2d]) / 1e3))
    steps = ["aacg_pipe_copy", "__amd_rocclr_fillBufferAligned", "aacg_parse_prepare", "aacg_parse_order_count", "aacg_parse_order_scan", "aacg_parse_order_fill", "aacg_parse_frames", "aacg_units_refresh", "aacg_imdct_run"]
    inwin = [r for r in kern if t0 <= int(r["Start_Timestamp"]) <= t1]
    print("  kernels of a batch (median begin -> end; a row includes its wait for CUs while other lanes' kernels run):")
    for s in steps:
        rs = [r for r in inwin if r["Kernel_Name"].startswith(s)]
        if not rs:
            continue
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        per = len(rs) / len(keep)
        name = rs[0]["Kernel_Name"] if s == "aacg_imdct_run" else s
        print("    %-34s %5.1f per batch   %8.1f us median   %8.1f max" % (name[:34], per, med(d) / 1e3, max(d) / 1e3))
    par = [r for r in inwin if r["Kernel_Name"].startswith("aacg_parse_frames")]
    print("  hardware queues the parse kernels ran on: %s" % sorted(set(int(r["Queue_Id"]) for r in par)))
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in par] + [(int(r["End_Timestamp"]), -1) for r in par])
    cur, last, acc = 0, t0, {}
    for t, d in ev:
        acc[cur] = acc.get(cur, 0) + (t - last)
        cur += d
        last = t
    tot = sum(acc.values()) or 1
    print("  parse kernels in flight at once: " + ", ".join("%d: %.0f %%" % (k, 100 * v / tot) for k, v in sorted(acc.items())))


if __name__ == "__main__":
    main()
