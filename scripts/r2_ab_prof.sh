#!/bin/bash
# per-kernel A/B of engine variants on the GPU box: rocprofv3 kernel stats of a 300-step march + adjoint sweep at 512^2
#   bash scripts/r2_ab_prof.sh <batch> <tag> ...      (tag "hip" = the default build, else libvch_<tag>.so)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
PK=$(ls -d sparse-optimal*_amd)
B=$1; shift
O=gpurun_out/r2_ab_prof.log
: > $O
for TAG in "$@"; do
  export VCH_LIB=$PWD/$PK/libvch_$TAG.so
  D=gpurun_out/abprof_$TAG
  rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python scripts/fwd_stats.py 512 300 $B > $D.out 2> $D.err || tail -3 $D.err
  echo "== $TAG (batch $B)" >> $O
  python - $(find $D -name "*kernel_stats.csv" | head -1) >> $O <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"])>0.8: print("%-44s %8s %8.2f us %6s %%"%(r["Name"].split("(")[0][:44],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
  rm -rf $D
done
cat $O
