#!/usr/bin/env python3
"""Config 5 (1024^2, amp = 1.0) march: per-step solver regime of trajectory 0 from the VCH_DEBUG_GUESS log
(spectral bound kappa_T, tolerances, sweeps, forms).  python scripts/r3_config5_probe.py [steps]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, vch_amd
    F2 = vch_amd.module("Vch_control_2D.Forward2_solver")
    N, M = 1024, int(sys.argv[2])
    e = vch_amd.Engine2D(Nx=N, Ny=N, batch=1, max_steps=M)
    phi0 = F2.init_phi_random(N, N, 1e-2, amp=1.0, seed=42)
    ph, st = e.forward(phi0, np.full(M, 1e-3), store=True)
    print("STATS", st, file=sys.stderr)
    for k in (0, 1, 2, 5, 10, 20, 50, M):
        if k <= M:
            a = np.abs(ph[k])
            print(f"LEVEL {k}: max|phi| {a.max():.4f}  share |phi| > 0.9: {np.mean(a > 0.9):.4f}  > 0.98: {np.mean(a > 0.98):.4f}", file=sys.stderr)
    sys.exit(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(M)], env=dict(os.environ, VCH_DEBUG_GUESS="1"),
                   capture_output=True, text=True)
step = 0
prev = 0
for ln in r.stderr.splitlines():
    if ln.startswith("guess order"):
        m = re.search(r"solves (\d+) sweeps (\d+) (\d+) (\d+) normR (\S+) .* tol (\S+) (\S+) (\S+) kT (\S+) (\S+) (\S+) form (\d) (\d) (\d) \| norms (\d+) total sweeps (\d+) newton (\d+) trials (\d+) lastform (\d) lastn (\d+)", ln)
        if m and (step < 12 or step % 10 == 0):
            sw = int(m.group(16))
            print(f"step {step:3d}: solves {m.group(1)} sweeps {m.group(2)} {m.group(3)} {m.group(4)} tol {m.group(6)} {m.group(7)} {m.group(8)} "
                  f"kT {float(m.group(9)):.3f} {float(m.group(10)):.3f} form {m.group(12)}{m.group(13)}{m.group(14)} | norms {m.group(15)} "
                  f"sweeps this step {sw - prev} trials so far {m.group(18)} normR {m.group(5)} last solve: form {m.group(19)} n {m.group(20)}")
        if m:
            prev = int(m.group(16))
        step += 1
    elif ln.startswith(("STATS", "LEVEL")):
        print(ln)
print(r.stderr[-1500:] if r.returncode else "")
