#!/bin/bash
# HBM traffic of the hot kernels from the TCC fabric counters (MI355X_MICROARCH.md §HBM):
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, each with --kernel-trace only.
# Run on the GPU box:  bash scripts/pmc_traffic.sh  ; results: gpurun_out/pmc/summary.json
set -e
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --time-steps 40 --contexts 1 --no-cpu-baseline --no-roofline"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/$CTR -- python bench.py $ARGS > $OUT/$CTR.json 2> $OUT/$CTR.err
done
python scripts/pmc_summarise.py $OUT > $OUT/summary.json
cp $OUT/summary.json gpurun_out/r2_pmc_traffic.json
cat $OUT/summary.json
find $OUT -name "*.csv" -size +20M -delete
