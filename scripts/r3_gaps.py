#!/usr/bin/env python3
"""Where does the time BETWEEN kernels go?  From a rocprofv3 --kernel-trace CSV of a single-stream run (one context): sorts
the dispatches by start time and reports, per kernel name, the idle gap in front of it (start - end of the previous dispatch),
plus the totals: kernel-busy time, idle time, the share of the idle time that sits in front of the first kernel of a time step
(the host's look at the state) and in front of everything else (dependent-launch latency).

    python scripts/r3_gaps.py <kernel_trace.csv> [--from-kernel k_eval] > summary.txt"""
import csv
import re
import statistics
import sys

path = sys.argv[1]
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        name = re.sub(r"\(.*", "", r.get("Kernel_Name") or r.get("Name")).replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
gaps, durs = {}, {}
busy = idle = 0.0
prev_end = rows[0][0]
for s, e, n in rows:
    g = (s - prev_end) * 1e-3
    d = (e - s) * 1e-3
    if g < 1e5:                      # not the pause between two marches
        gaps.setdefault(n, []).append(g)
        idle += max(g, 0.0)
    durs.setdefault(n, []).append(d)
    busy += d
    prev_end = max(prev_end, e)
span = (rows[-1][1] - rows[0][0]) * 1e-3
print(f"# {len(rows)} dispatches, span {span * 1e-3:.1f} ms, kernel time {busy * 1e-3:.1f} ms, idle between kernels {idle * 1e-3:.1f} ms")
print(f"{'kernel':44s} {'calls':>7s} {'gap_mean_us':>11s} {'gap_med_us':>10s} {'gap_total_ms':>12s} {'dur_mean_us':>11s} {'dur_total_ms':>12s}")
for n, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n[:44]:44s} {len(v):7d} {sum(v) / len(v):11.2f} {statistics.median(v):10.2f} {sum(v) * 1e-3:12.2f} "
          f"{sum(durs[n]) / len(durs[n]):11.2f} {sum(durs[n]) * 1e-3:12.2f}")
