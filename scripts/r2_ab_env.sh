#!/bin/bash
# A/B of run-time knobs (environment variables) on the GPU box: parity subset + march / adjoint time at 512^2 x 1000
#   bash scripts/r2_ab_env.sh "VAR=val" "VAR=val" ...     ("" = defaults)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r2_ab_env.log
: > $O
for KV in "$@"; do
  echo "== ${KV:-defaults}" >> $O
  env $KV python -m pytest tests/test_gpu_2d.py -q -x -k "spectral or forward_backward" 2>&1 | tail -1 >> $O
  for B in 8 4 1; do env $KV python scripts/fwd_stats.py 512 1000 $B 2>&1 | grep -E "^forward|^backward" | tail -2 | sed "s/^/B=$B /" | cut -c1-20,240-330 >> $O; done
done
cat $O
