#!/usr/bin/env python3
"""Where a DCT row-pass workgroup spends its life (tuning build -DVCH_FFT_TIMING, GPU box):
per-workgroup s_memrealtime stamps {start, image loaded, FFT done, stored}, 512^2 x B."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, ".")
import vch_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = vch_amd.Engine2D(Nx=512, Ny=512, batch=B, max_steps=2)
e.spectral_solve(1.0, 0.1, 1e-3, np.random.default_rng(0).standard_normal((B, 513, 513)))   # fills tmp[0], warms up
n = B * 257
buf = (C.c_longlong * (4 * n))()
f = e.lib.vch2d_debug_fft_phases
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
got = f(e.ctx, buf, 4 * n)
assert got == n, got
t = np.frombuffer(buf, dtype=np.int64).reshape(n, 4).astype(np.float64) * 0.01      # us (100 MHz)
t0 = t[:, 0].min()
ph = np.diff(t, axis=1)
print(f"B={B}: workgroups {n}; kernel span {t[:, 3].max() - t0:.2f} us; start spread {t[:, 0].max() - t0:.2f} us")
print("per-workgroup phase (us)  mean / p10 / p90:")
for k, name in enumerate(("load image", "FFT (5 passes)", "store")):
    print(f"  {name:16s} {ph[:, k].mean():6.2f} {np.percentile(ph[:, k], 10):6.2f} {np.percentile(ph[:, k], 90):6.2f}")
print(f"  lifetime         {(t[:, 3] - t[:, 0]).mean():6.2f}")
