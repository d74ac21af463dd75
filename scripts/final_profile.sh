#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel stats of the bench command + PMC traffic.
set -e
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/final
mkdir -p $OUT
rm -rf $OUT/stats
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || tail -5 $OUT/bench_under_rocprof.err
find $OUT/stats -name "*kernel_trace*" -delete
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -25 $OUT/kernel_stats.csv | cut -c1-160
tail -c 1500 $OUT/bench_under_rocprof.json
