#!/bin/bash
# 1D evidence on the GPU box: rocprofv3 kernel stats of config 2 (N = 4096, 1000 steps) at batch 1
# and batch 256, and of config 1 (N = 256, 200 steps) at batch 256; summaries -> gpurun_out/r2prof1d/
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r2prof1d
mkdir -p $OUT
run() {  # tag N steps dt batch
  rm -rf $OUT/st_$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$1 -- python scripts/bench_1d.py $2 $3 $4 $5 > $OUT/bench_$1.json 2> $OUT/bench_$1.err || tail -3 $OUT/bench_$1.err
  find $OUT/st_$1 -name "*kernel_trace*" -delete
  cp $(find $OUT/st_$1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$1.csv
  rm -rf $OUT/st_$1
  echo "== $1"; head -5 $OUT/kernel_stats_$1.csv | cut -c1-60,100-220; cat $OUT/bench_$1.json
}
run n4096_b1 4096 1000 1e-3 1
run n4096_b256 4096 1000 1e-3 256
run n256_b256 256 200 5e-3 256
