#!/usr/bin/env python3
"""Residual norms after the first / second Newton iteration of every step (trajectory 0) along PGD-controlled marches, and
the sweeps of the first solve: how much slack is there between the first solve's tolerance and what the Newton iteration
needs?  python scripts/r3_r1_stats.py [N] [M] [B] [iters]   (GPU box; VCH_DEBUG_GUESS log of scripts/r3_cheb_stats.py --child)"""
import os, re, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, M, B, iters = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 512), (2, 1000), (3, 4), (4, 2)))
r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r3_cheb_stats.py"), "--child", str(N), str(M), str(B), str(iters)],
                   env=dict(os.environ, VCH_DEBUG_GUESS="1"), capture_output=True, text=True)
pat = re.compile(r"solves (\d+) sweeps (\d+) (\d+) .* tol (\S+) (\S+) .* R (\S+) (\S+) (\S+) r0")
seg = []
k = 0
for ln in r.stderr.splitlines():
    m = pat.search(ln)
    if not m:
        continue
    seg.append((int(m.group(1)), int(m.group(2)), float(m.group(4)), float(m.group(6)), float(m.group(7)), float(m.group(8))))
    if len(seg) == M:
        two = [s for s in seg if s[0] >= 2]
        q = lambda v, f: sorted(v)[min(len(v) - 1, int(f * len(v)))]
        R1 = [s[4] for s in two]
        print(f"march {k}: steps with >= 2 solves {len(two)} of {M}; R_0 median {statistics.median(s[3] for s in seg):.2e}; "
              f"R_1 5/50/95 %: {q(R1, .05):.2e} {q(R1, .5):.2e} {q(R1, .95):.2e}; first-solve tol median {statistics.median(s[2] for s in seg):.2e}; "
              f"first-solve sweeps mean {statistics.mean(s[1] for s in seg):.2f}; R_1 of 1-solve steps max {max([s[4] for s in seg if s[0] == 1] or [0]):.2e}")
        # smoothness of R_1 from step to step
        ratios = [two[i + 1][4] / two[i][4] for i in range(len(two) - 1) if two[i][4] > 0]
        if ratios:
            print(f"          R_1 step-to-step ratio 1/50/99 %: {q(ratios, .01):.2f} {q(ratios, .5):.2f} {q(ratios, .99):.2f}")
        seg = []
        k += 1
print(r.stderr[-800:] if r.returncode else "")
