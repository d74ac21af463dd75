#!/bin/bash
# rocprofv3 kernel trace of a 512^2 x 1000 march at B trajectories in ONE context, summarised as inter-kernel gaps
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
OUT=gpurun_out/r3gaps
mkdir -p $OUT
rm -rf $OUT/t_$B
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$B -- python scripts/fwd_stats.py 512 1000 $B > $OUT/march_$B.txt 2>&1 || tail -5 $OUT/march_$B.txt
TR=$(find $OUT/t_$B -name "*kernel_trace.csv" | head -1)
python scripts/r3_gaps.py $TR > $OUT/gaps_b$B.txt
rm -rf $OUT/t_$B
head -40 $OUT/gaps_b$B.txt
