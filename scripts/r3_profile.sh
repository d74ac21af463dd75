#!/bin/bash
# Round-3 evidence on the GPU box: rocprofv3 kernel trace of the bench command (no in-bench event pairs), summarised per
# kernel over live launches (scripts/r3_live_stats.py) + rocprofv3's own stats table.  $1 = tag, rest = bench arguments.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}; shift
OUT=gpurun_out/r3prof
mkdir -p $OUT
rm -rf $OUT/stats_$TAG
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || tail -5 $OUT/bench_$TAG.err
TR=$(find $OUT/stats_$TAG -name "*kernel_trace.csv" | head -1)
python scripts/r3_live_stats.py $TR > $OUT/live_stats_$TAG.txt
cp $(find $OUT/stats_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$TAG.csv
rm -rf $OUT/stats_$TAG
head -30 $OUT/live_stats_$TAG.txt
