#!/bin/bash
# A/B of engine variants on the GPU box: correctness (spectral solve + forward golden tests) and march time
#   bash scripts/r2_ab.sh <tag> ...      (tag "hip" = the default build, else libvch_<tag>.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
PK=$(ls -d sparse-optimal*_amd)
O=gpurun_out/r2_ab.log
: > $O
for TAG in "$@"; do
  export VCH_LIB=$PWD/$PK/libvch_$TAG.so
  echo "== $TAG" >> $O
  python -m pytest tests/test_gpu_2d.py -q -x -k "spectral or forward_backward" 2>&1 | tail -1 >> $O
  for B in 8 4 1; do python scripts/fwd_stats.py 512 1000 $B 2>&1 | grep -E "^forward|^backward" | tail -2 | sed "s/^/B=$B /" | cut -c1-20,240-330 >> $O; done
done
cat $O
