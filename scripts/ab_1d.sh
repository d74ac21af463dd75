#!/bin/bash
# A/B of prebuilt engine variants on the 1D configs:  bash scripts/ab_1d.sh hip t512 t1024
cd ${GRAFT_REPO_ROOT:-/root/repo}
PK=$(ls -d sparse-optimal*_amd)
for TAG in "$@"; do
  export VCH_LIB=$PWD/$PK/libvch_$TAG.so
  echo "== $TAG"
  timeout -k 10 200 python scripts/bench_1d.py 4096 1000 1e-3 1 || break
  timeout -k 10 200 python scripts/bench_1d.py 4096 1000 1e-3 256 || break
  timeout -k 10 200 python scripts/bench_1d.py 256 200 5e-3 256 || break
  timeout -k 10 200 python -m pytest tests/test_gpu_1d.py -m gpu -q -x 2>&1 | tail -2
done
