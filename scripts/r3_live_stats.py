#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV over LIVE launches only: a launch whose trajectories are all
gated off (the state machine has nothing for it to do) returns in about a microsecond and would flatter the mean, so
dispatches shorter than --noop-us (default 3) are counted separately.

    python scripts/r3_live_stats.py <kernel_trace.csv> [--noop-us 3] > summary.txt
Columns: calls, live calls, no-op share of the calls, mean / median live duration (us), share of the total kernel time."""
import csv
import re
import statistics
import sys

path = sys.argv[1]
thr = 3.0
if "--noop-us" in sys.argv:
    thr = float(sys.argv[sys.argv.index("--noop-us") + 1])
per = {}
with open(path, newline="") as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        name = r.get("Kernel_Name") or r.get("Name")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        per.setdefault(name, []).append(d)
tot = sum(sum(v) for v in per.values())
print(f"# {path}: {sum(len(v) for v in per.values())} dispatches, {tot * 1e-3:.1f} ms of kernel time; no-op = shorter than {thr} us")
print(f"{'kernel':48s} {'calls':>8s} {'live':>8s} {'noop%':>6s} {'live_mean_us':>12s} {'live_med_us':>11s} {'share%':>7s}")
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    live = [x for x in v if x >= thr]
    print(f"{name[:48]:48s} {len(v):8d} {len(live):8d} {100.0 * (1 - len(live) / len(v)):6.1f} "
          f"{(sum(live) / len(live) if live else 0.0):12.2f} {(statistics.median(live) if live else 0.0):11.2f} {100.0 * sum(v) / tot:7.2f}")
