#!/bin/bash
# SQ wave-time breakdown of the sweep kernels (what bounds them?).  GPU box only.
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/pmc_sq
rm -rf $OUT && mkdir -p $OUT
pass() {  # tag counters...
  tag=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python bench.py --steps 1 --warmup 0 --time-steps 20 --no-cpu-baseline --no-roofline --contexts 1 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || tail -5 $OUT/bench_$tag.err
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("gpurun_out/pmc_sq/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        n=r["Kernel_Name"]
        key=next((k for k in ("k_dct_rows<4","k_dct_rows<3","k_dct_rows<0","k_dct_rows<5","k_dct_cols","k_cg_rows_fwd<0","k_cg_rows_fwd<1","k_adj_rows_fwd<0","k_residual<1","k_dmu_ceiling_fin") if k in n),None)
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in sorted(acc.items()):
    print(k)
    for c,v in sorted(d.items()):
        vm=max(v); live=[x for x in v if x>0.25*vm] or v
        print("   %-26s mean %.4g  (n=%d of %d)"%(c,sum(live)/len(live),len(live),len(v)))
PY
find $OUT -name "*.csv" -size +10M -delete
