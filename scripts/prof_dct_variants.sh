set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in 1024 2048 4096; do
  export VCH_DCT_COLS_C=$C
  rm -rf gpurun_out/prof_$C
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$C -- python bench.py --steps 1 --warmup 0 --time-steps 100 --no-cpu-baseline --no-roofline > gpurun_out/prof_$C.json 2> gpurun_out/prof_$C.err
  find gpurun_out/prof_$C -name "*kernel_trace*" -delete
  echo "== cols C=$C"; python - <<PY
import json,glob,csv
print(json.load(open("gpurun_out/prof_$C.json"))["ms_per_step"])
f=glob.glob("gpurun_out/prof_$C/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"], round(float(r["AverageNs"])/1000,1), r["Percentage"])
PY
done
