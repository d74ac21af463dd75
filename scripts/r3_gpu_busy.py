#!/usr/bin/env python3
"""GFX-busy share (sysfs gpu_busy_percent, sampled every 20 ms) while bench.py runs its timed region.
    python scripts/r3_gpu_busy.py [bench args ...]      (GPU box)"""
import glob
import json
import subprocess
import sys
import threading
import time

paths = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
print("sysfs files:", paths, flush=True)
samples = []
stop = [False]


def poll():
    while not stop[0]:
        t = time.time()
        row = []
        for p in paths:
            try:
                row.append(int(open(p).read()))
            except Exception:
                row.append(-1)
        samples.append((t, row))
        time.sleep(0.02)


th = threading.Thread(target=poll, daemon=True)
th.start()
t0 = time.time()
r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-roofline"] + sys.argv[1:], capture_output=True, text=True)
t1 = time.time()
stop[0] = True
th.join()
line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
print(r.stderr[-2000:] if r.returncode else "", flush=True)
d = json.loads(line)
timed = d["ms_per_step"] * d["steps"] * 1e-3
print(f"value {d['value']:.3f}, timed region {timed:.1f} s (the last {timed:.1f} s of the {t1 - t0:.1f} s run)")
# the timed region ends just before the process prints and exits: take the window [t1 - 1.0 - timed, t1 - 1.0]
for i, p in enumerate(paths):
    w = [row[i] for t, row in samples if t1 - 1.0 - timed <= t <= t1 - 1.0 and row[i] >= 0]
    if w and max(w) > 0:
        w.sort()
        print(f"{p}: {len(w)} samples, mean {sum(w) / len(w):.1f} %, 5/50/95 %: {w[len(w) // 20]} {w[len(w) // 2]} {w[-len(w) // 20 - 1]}")
