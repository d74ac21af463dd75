cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in hip fakelog; do
  export VCH_LIB=$PWD/sparse-optimal-control-of-viscous-chan-hilliard-via-gradient-descent--1d-2d_amd/libvch_$v.so
  rm -rf gpurun_out/fk_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fk_$v -- python scripts/fwd_stats.py 512 60 8 > gpurun_out/fk_$v.txt 2>&1
  python scripts/r3_live_stats.py $(find gpurun_out/fk_$v -name "*kernel_trace.csv" | head -1) | head -8 > gpurun_out/fk_live_$v.txt
  rm -rf gpurun_out/fk_$v
done
