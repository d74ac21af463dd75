#!/bin/bash
# idle time between consecutive kernels of one context (GPU box): rocprofv3 kernel trace of a 200-step march + adjoint
# sweep at 512^2, batch $1; gap = start(k+1) - end(k) on the device timeline, summed per preceding kernel
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
D=gpurun_out/gaps
rm -rf $D
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python scripts/fwd_stats.py 512 200 $B > $D.out 2> $D.err || tail -3 $D.err
python - $(find $D -name "*kernel_trace.csv" | head -1) > gpurun_out/r2_gaps_b$B.txt <<'PY'
import csv,sys,collections
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
busy=sum(e-s for s,e,_ in rows); span=rows[-1][1]-rows[0][0]
print("kernels %d  busy %.1f ms  span %.1f ms"%(len(rows),busy/1e6,span/1e6))
g=collections.defaultdict(list)
for (s0,e0,n0),(s1,e1,n1) in zip(rows,rows[1:]):
    g[n0+" -> "+n1].append(s1-e0)
tot=sum(sum(v) for v in g.values())
print("total gap %.1f ms"%(tot/1e6))
hist=collections.Counter()
for v in g.values():
    for x in v: hist[min(int(x/1000),50)]+=1
print("gap histogram (us: count):", sorted(hist.items()))
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:40]:
    v2=sorted(v)
    print("%-70s n %6d  mean %7.2f us  median %7.2f  p90 %7.2f  total %7.2f ms"%(k[:70],len(v),sum(v)/len(v)/1e3,v2[len(v)//2]/1e3,v2[int(len(v)*0.9)]/1e3,sum(v)/1e6))
PY
rm -rf $D
head -60 gpurun_out/r2_gaps_b$B.txt
