#!/bin/bash
# A/B of run-time switches of one build:  bash scripts/ab_env.sh "TAG1:VAR=VAL" "TAG2:" ...  (bench, 100 steps, 1 context, rocprof stats)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
for SPEC in "$@"; do
  TAG=${SPEC%%:*}; KV=${SPEC#*:}
  if [ -n "$KV" ]; then export "$KV"; fi
  rm -rf gpurun_out/ab/$TAG
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab/$TAG -- python bench.py --steps 1 --warmup 0 --time-steps 100 --no-cpu-baseline --no-roofline --contexts 1 > gpurun_out/ab/$TAG.json 2> gpurun_out/ab/$TAG.err || { tail -5 gpurun_out/ab/$TAG.err; }
  if [ -n "$KV" ]; then unset "${KV%%=*}"; fi
  find gpurun_out/ab/$TAG -name "*kernel_trace*" -delete
  echo "== $TAG"; python - <<PY
import json,glob,csv
d=json.load(open("gpurun_out/ab/$TAG.json")); print("ms_per_step", round(d["ms_per_step"],1), "cost", d["cost_sum_per_step"])
f=glob.glob("gpurun_out/ab/$TAG/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:58].ljust(58), r["Calls"], round(float(r["AverageNs"])/1000,1), r["Percentage"])
PY
done
