#!/bin/bash
# column-pass width A/B on the GPU box (VCH_COLS_C = complex image per workgroup: 1024 / 2048 / 4096 = 2 / 4 / 8 columns)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_cols.log
: > $O
for CC in 1024 2048 4096; do
  echo "== VCH_COLS_C=$CC" >> $O
  VCH_COLS_C=$CC python -m pytest tests/test_gpu_2d.py -q -x -k "spectral or forward_backward" 2>&1 | tail -1 >> $O
  for B in 8 4 1; do VCH_COLS_C=$CC python scripts/fwd_stats.py 512 1000 $B 2>&1 | grep -E "^forward|^backward" | tail -2 | sed "s/^/B=$B /" | cut -c1-20,240-330 >> $O; done
done
cat $O
