#!/bin/bash
# round-2 probes on the GPU box: march statistics under the schedule / tolerance knobs and at small batches
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r2_probe.log
: > $O
run() { echo "== $*" >> $O; env "$@" python scripts/fwd_stats.py 512 1000 ${B:-8} >> $O 2>&1; }
B=8 run VCH_X=1
B=8 run VCH_LIN_ETA=0
B=8 run VCH_LIN_ETA=2e-7
B=8 run VCH_NO_SPEC=1
B=4 run VCH_X=1
B=2 run VCH_X=1
B=1 run VCH_X=1
tail -40 $O
