#!/bin/bash
# rocprofv3 kernel trace of one uncontrolled 512^2 x 1000 march + adjoint sweep at 8 trajectories per launch
# (scripts/fwd_stats.py), summarised over live launches.  $1 = tag; environment variables select the build variant.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-march}; B=${2:-8}
OUT=gpurun_out/r3prof
mkdir -p $OUT
rm -rf $OUT/m_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/m_$TAG -- python scripts/fwd_stats.py 512 1000 $B > $OUT/march_$TAG.txt 2>&1 || tail -5 $OUT/march_$TAG.txt
TR=$(find $OUT/m_$TAG -name "*kernel_trace.csv" | head -1)
python scripts/r3_live_stats.py $TR > $OUT/march_live_$TAG.txt
rm -rf $OUT/m_$TAG
head -32 $OUT/march_live_$TAG.txt
