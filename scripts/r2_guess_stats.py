#!/usr/bin/env python3
"""Statistics of the starting guesses from a VCH_DEBUG_GUESS=1 log (stderr of bench.py / fwd_stats.py): per march segment,
orders chosen, sweeps of the first and second solve, median ratios."""
import re, sys, collections, statistics
L = [l for l in open(sys.argv[1]) if l.startswith("guess order")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
print(len(L), "steps")
for k in range(0, len(L), M):
    seg = L[k:k + M]
    o1, o2, s1, s2, r1, r2 = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter(), [], []
    for l in seg:
        m = re.search(r"order (\d+) / (\d+) \(run (\d+)\) \| traj 0: ratio (\S+) / (\S+) solves (\d) sweeps (\d+) (\d+)", l)
        if not m or int(m.group(6)) == 0:
            continue
        o1[int(m.group(1))] += 1; o2[int(m.group(2))] += 1; s1[int(m.group(7))] += 1; s2[int(m.group(8))] += 1
        r1.append(float(m.group(4))); r2.append(float(m.group(5)))
    if r1:
        print("march %2d: orders1 %s sweeps1 %s | orders2 %s sweeps2 %s | median ratio1 %.1e ratio2 %.1e" % (
            k // M, sorted(o1.items()), sorted(s1.items()), sorted(o2.items()), sorted(s2.items()), statistics.median(r1),
            statistics.median([x for x in r2 if x > 0] or [0])))
