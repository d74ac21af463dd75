#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (mean per dispatch over the
non-skipped dispatches) and compare with the algorithmic bytes of DESIGN.md §4.

Counter units/corrections as prescribed by MI355X_MICROARCH.md §HBM: the counters are in KiB
(x1024); on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x for
16 B/lane accesses and is uncalibrated for other widths, so the 8 B/lane read pattern of these
kernels is calibrated in the same run on k_grad_prox, whose reads (u, r: 16 B/node, streamed once
from histories far larger than the Infinity Cache) are known exactly."""
import csv, glob, json, sys, collections

out = sys.argv[1]
N, B, M = 512, 8, 40
nodes = (N + 1) * (N + 1) * B
# (read, write) algorithmic bytes per node per dispatch; third entry: which read calibration applies
#   "stream": 8 B/lane coalesced streaming reads (row FFT ingest, element-wise kernels), calibrated on k_grad_prox
#   "tile":   8 B/lane reads of 64 x 16 tiles (the stencil kernels), calibrated on k_copy_plane (a pure tiled copy)
alg = {"k_cheb_rows<1024, 10, 0>": (40, 16, "stream"), "k_cheb_rows<1024, 10, 1>": (16, 24, "stream"),
       "k_eval<2, false>": (56, 40, "tile"), "k_eval<0, false>": (40, 64, "tile"),
       "k_cg_rows_fwd<0, 1024, 10>": (40, 32, "stream"), "k_cg_rows_fwd<1, 1024, 10>": (16, 24, "stream"),
       "k_dct_rows<0, 1024, 10>": (8, 8, "stream"), "k_dct_rows<3, 1024, 10>": (16, 8, "stream"),
       "k_dct_rows<4, 1024, 10>": (24, 8, "stream"), "k_dct_rows<5, 1024, 10>": (24, 8, "stream"),
       "k_dct_cols<1024, 10>": (8, 8, "stream"), "k_adj_rows_fwd<0, 1024, 10>": (40, 32, "stream"),
       "k_residual<1>": (48, 40, "tile"), "k_residual<0>": (32, 32, "tile"), "k_prepare": (40, 32, "tile"),
       "k_dmu_ceiling_fin": (40, 16, "tile"), "k_adj_rhs": (48, 16, "tile"), "k_schur_p<0>": (40, 32, "tile"),
       "k_copy_plane": (8, 8, "tile"), "k_grad_prox": (16, 8, "stream"), "k_cost": (24, 0, "stream"),
       # starting guesses at their full order (six increments / four levels; the first steps of a march use fewer planes)
       "k_guess": (64, 16, "tile"), "k_adj_guess": (40, 16, "tile")}
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{out}/{ctr}/*/*counter_collection.csv")
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            name = r["Kernel_Name"]
            key = next((k for k in alg if name.startswith("void " + k) or name.startswith(k)), None)
            if key:
                acc[key].append(float(r["Counter_Value"]) * 1024.0)
    for k, v in acc.items():
        vmax = max(v)
        live = [x for x in v if x > 0.25 * vmax]          # drop gated (skipped) dispatches
        res.setdefault(k, {})[ctr] = dict(dispatches=len(v), live=len(live), mean_bytes=sum(live) / max(len(live), 1))
summary = {}
levels = M + 1
cal = {"stream": None, "tile": None}
if "k_grad_prox" in res and "FETCH_SIZE" in res["k_grad_prox"]:
    cal["stream"] = res["k_grad_prox"]["FETCH_SIZE"]["mean_bytes"] / (16.0 * nodes * levels)
if "k_copy_plane" in res and "FETCH_SIZE" in res["k_copy_plane"]:
    cal["tile"] = res["k_copy_plane"]["FETCH_SIZE"]["mean_bytes"] / (8.0 * nodes)
for k, d in res.items():
    rd, wr, kind = alg[k]
    mult = levels if k in ("k_grad_prox", "k_cost") else 1
    f = d.get("FETCH_SIZE", {}).get("mean_bytes")
    w = d.get("WRITE_SIZE", {}).get("mean_bytes")
    c = cal[kind] or cal["stream"]
    summary[k] = dict(algorithmic_read=rd * nodes * mult, algorithmic_write=wr * nodes * mult,
                      fetch_raw=f, write_raw=w, read_calibration=kind,
                      fetch_corrected=(f / c if (f is not None and c) else None),
                      traffic_over_algorithmic=(((f / c if c else f) + w) / ((rd + wr) * nodes * mult)
                                                if (f is not None and w is not None) else None),
                      dispatches=d.get("FETCH_SIZE", d.get("WRITE_SIZE"))["dispatches"])
print(json.dumps(dict(trajectories_per_launch=B, grid=N, calibration_fetch_ratio=cal, kernels=summary), indent=1))
