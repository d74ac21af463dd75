#!/bin/bash
# Round-2 evidence on the GPU box: rocprofv3 kernel stats of the bench command (no in-bench event pairs: --no-roofline),
# once with the default two contexts and once with one context of 8; summaries -> gpurun_out/r2prof/
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r2prof
mkdir -p $OUT
for K in 2 1; do
  rm -rf $OUT/stats_c$K
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$K -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --contexts $K > $OUT/bench_c$K.json 2> $OUT/bench_c$K.err || tail -5 $OUT/bench_c$K.err
  find $OUT/stats_c$K -name "*kernel_trace*" -delete
  cp $(find $OUT/stats_c$K -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c$K.csv
  rm -rf $OUT/stats_c$K
  echo "== contexts $K"; head -16 $OUT/kernel_stats_c$K.csv | cut -c1-70,170-260
done
