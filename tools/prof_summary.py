#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        lines.append("  %-40s calls %6s  avg_ns %10s  min %10s  max %10s  total_ns %12s  %%%s" % (
            row.get("Name", "")[:40], row.get("Calls"), row.get("AverageNs"), row.get("MinNs"), row.get("MaxNs"),
            row.get("TotalDurationNs"), row.get("Percentage")))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if not k.startswith("aacg_"):
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    lines.append("== PMC per dispatch (mean over dispatches): %s" % k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        lines.append("  %-28s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
text = "\n".join(lines)
open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
print(text)
