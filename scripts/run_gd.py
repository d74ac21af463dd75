#!/usr/bin/env python3
"""Run the PGD driver non-interactively on the GPU, as `python GD2_configured.py` / `python GD_1D.py` do in the
reference (minus prompts and plots):   python scripts/run_gd.py {1d|2d} [--iters N] [--params FILE]"""
import argparse, sys
sys.path.insert(0, ".")
import vch_amd

ap = argparse.ArgumentParser()
ap.add_argument("dim", choices=["1d", "2d"])
ap.add_argument("--iters", type=int, default=None, help="iterations (default: max_iter of the configuration)")
ap.add_argument("--params", default=None, help="last-run JSON (default: the reference's file name in the cwd)")
a = ap.parse_args()
vch_amd.build()
if a.dim == "2d":
    G = vch_amd.module("Vch_control_2D.GD2_configured")
    G.main(n_iter=a.iters, **({"params_file": a.params} if a.params else {}))
else:
    G = vch_amd.module("Vch_control_1D.GD_1D")
    G.main(n_iter=a.iters, **({"params_file": a.params} if a.params else {}))
