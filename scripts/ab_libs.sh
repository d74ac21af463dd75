#!/bin/bash
# A/B of prebuilt engine variants (…_amd/libvch_<tag>.so) on the GPU box:
#   bash scripts/ab_libs.sh hip p1d0 p0d1 ...
set -e
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
PK=$(ls -d sparse-optimal*_amd)
mkdir -p gpurun_out/ab
for TAG in "$@"; do
  export VCH_LIB=$PWD/$PK/libvch_$TAG.so
  rm -rf gpurun_out/ab/$TAG
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/$TAG -- python bench.py --steps 1 --warmup 0 --time-steps 100 --no-cpu-baseline --no-roofline --contexts 1 > gpurun_out/ab/$TAG.json 2> gpurun_out/ab/$TAG.err || { tail -5 gpurun_out/ab/$TAG.err; continue; }
  find gpurun_out/ab/$TAG -name "*kernel_trace*" -delete
  echo "== $TAG"; python - <<PY
import json,glob,csv
d=json.load(open("gpurun_out/ab/$TAG.json")); print("ms_per_step", round(d["ms_per_step"],1), "cost", d["cost_sum_per_step"])
f=glob.glob("gpurun_out/ab/$TAG/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print(r["Name"][:58].ljust(58), r["Calls"], round(float(r["AverageNs"])/1000,1), r["Percentage"])
PY
done
