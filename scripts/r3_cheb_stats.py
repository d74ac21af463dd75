#!/usr/bin/env python3
"""What would a reduction-free (Chebyshev) sweep schedule cost?  Runs PGD iterations at the headline size with
VCH_DEBUG_GUESS=1 in a child process (one line per time step on stderr: tolerance, spectral bound kappa_T and CG sweeps of
the first three solves of trajectory 0) and prices, for every solve, the number of Chebyshev sweeps the rigorous bound
asks for: the smallest n with 1 / T_{n+1}(sigma) <= tol, sigma = (kappa_T + 1) / (kappa_T - 1) (the preconditioner
application that starts the solve already gives the degree-1 polynomial).

    python scripts/r3_cheb_stats.py [N] [M] [B] [iters]      (GPU box)
"""
import collections
import math
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(N, M, B, iters):
    sys.path.insert(0, ROOT)
    import numpy as np
    import vch_amd
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
    t, dts = vch_amd.time_grid(1.0 * M / 1000.0, 1e-3)
    eng = vch_amd.Engine2D(Nx=N, Ny=N, batch=B, max_steps=len(dts))
    phi0 = np.stack([F2.init_phi_random(N, N, 1e-2, amp=0.1, seed=42 + i) for i in range(B)])
    xs = np.linspace(0, 1, N + 1)
    phi_T = np.broadcast_to(0.7 * np.sin(2 * np.pi * xs)[:, None] * np.cos(np.pi * xs)[None, :], phi0.shape).copy()
    print("MARCH init", file=sys.stderr, flush=True)
    eng.pgd_init(phi0, phi_T, t, vch_amd.make_opt(), ramp=True, T=1.0 * M / 1000.0)
    for k in range(iters):
        print(f"MARCH iter {k}", file=sys.stderr, flush=True)
        out = eng.pgd_iterate(1)
        print("attempts", out["attempts"][:, 0], "cost", out["cost"][:, 0], file=sys.stderr, flush=True)
    eng.close()


def n_cheb(tol, kT):
    if not (kT > 1.0):
        return 0
    sigma = (kT + 1.0) / (kT - 1.0)
    a = math.acosh(sigma)
    n = 0
    while n < 200 and 1.0 / math.cosh((n + 1) * a) > tol:
        n += 1
    return n


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(*[int(v) for v in sys.argv[2:6]])
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    env = dict(os.environ, VCH_DEBUG_GUESS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(N), str(M), str(B), str(iters)], env=env,
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-4000:])
        sys.exit(r.returncode)
    pat = re.compile(r"solves (\d+) sweeps (\d+) (\d+) (\d+) .* tol (\S+) (\S+) (\S+) kT (\S+) (\S+) (\S+)")
    seg, name, count = [], "?", [0]

    def flush():
        if not seg:
            return
        cg = collections.Counter()
        ch = collections.Counter()
        tot_cg = tot_ch = solves = 0
        tols, kts = [], []
        for ns, sw, tol, kt in seg:
            for j in range(min(ns, 3)):
                solves += 1
                n = n_cheb(tol[j], kt[j])
                cg[(j, sw[j])] += 1
                ch[(j, n)] += 1
                tot_cg += sw[j]
                tot_ch += n
                tols.append(tol[j])
                kts.append(kt[j])
        tols.sort()
        kts.sort()
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
        print(f"{name}: steps {len(seg)} solves {solves} | CG sweeps/solve {tot_cg / max(solves, 1):.2f}  Chebyshev sweeps/solve "
              f"{tot_ch / max(solves, 1):.2f}")
        print(f"   tol quantiles 5/50/95 %: {q(tols, .05):.1e} {q(tols, .5):.1e} {q(tols, .95):.1e}   kappa_T 5/50/95 %: "
              f"{q(kts, .05):.4f} {q(kts, .5):.4f} {q(kts, .95):.4f}")
        for j in range(3):
            a = sorted((k[1], v) for k, v in cg.items() if k[0] == j)
            b = sorted((k[1], v) for k, v in ch.items() if k[0] == j)
            if a:
                print(f"   solve {j + 1}: CG sweeps {a}  Chebyshev sweeps {b}")
    for ln in r.stderr.splitlines():
        if ln.startswith("MARCH") or ln.startswith("attempts"):
            if ln.startswith("attempts"):
                print("  ", ln)
            continue
        if ln.startswith("adjoint level"):
            continue
        m = pat.search(ln)
        if not m:
            continue
        seg.append((int(m.group(1)), [int(m.group(k)) for k in (2, 3, 4)], [float(m.group(k)) for k in (5, 6, 7)],
                    [float(m.group(k)) for k in (8, 9, 10)]))
        if len(seg) == M:
            name = f"march {count[0]}"
            flush()
            count[0] += 1
            seg.clear()
    name = "tail"
    flush()


if __name__ == "__main__":
    main()
