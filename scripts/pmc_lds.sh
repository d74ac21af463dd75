#!/bin/bash
# SQ counters of the DCT kernels (is the FFT LDS-bound?).  GPU box only.
set -e
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/pmc_lds
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
grep -c . $OUT/counters_list.txt
grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" $OUT/counters_list.txt | sort -u | tr '\n' ' '; echo
CTRS="${1:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES}"
timeout -k 10 400 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/run -- python bench.py --steps 1 --warmup 0 --time-steps 20 --no-cpu-baseline --no-roofline --contexts 1 > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_lds/run/*/*counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        n=r["Kernel_Name"]
        key=next((k for k in ("k_dct_rows<0","k_dct_rows<3","k_dct_cols","k_schur_p<0","k_cg_update<0") if k in n),None)
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        vm=max(v); live=[x for x in v if x>0.25*vm] or v
        print("   %-26s mean %.4g  (n=%d)"%(c,sum(live)/len(live),len(live)))
PY
find $OUT -name "*.csv" -size +10M -delete
