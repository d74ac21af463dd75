#!/bin/bash
# Round-3 evidence on the GPU box, final build: rocprofv3 live-launch summaries of the bench command (2 and 1 contexts) and of
# one march + sweep at 8 trajectories alone on the chip, PMC traffic, then the bench lines (which attach those summaries).
# Everything lands under gpurun_out/r3final/; the summaries are also copied into profiles/ ON THE BOX before the bench runs.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3final
mkdir -p $OUT
bash scripts/r3_profile.sh c2 --contexts 2 > $OUT/prof_c2.log 2>&1
bash scripts/r3_profile.sh c1 --contexts 1 > $OUT/prof_c1.log 2>&1
bash scripts/r3_prof_march.sh b8 8 > $OUT/prof_march_b8.log 2>&1
cp gpurun_out/r3prof/live_stats_c2.txt profiles/live_stats_contexts2.txt
cp gpurun_out/r3prof/live_stats_c1.txt profiles/live_stats_contexts1.txt
cp gpurun_out/r3prof/kernel_stats_c2.csv profiles/kernel_stats_contexts2.csv
cp gpurun_out/r3prof/kernel_stats_c1.csv profiles/kernel_stats_contexts1.csv
cp gpurun_out/r3prof/march_live_b8.txt profiles/live_stats_march_b8.txt
cp profiles/live_stats_contexts2.txt profiles/live_stats_contexts1.txt profiles/kernel_stats_contexts2.csv profiles/kernel_stats_contexts1.csv profiles/live_stats_march_b8.txt gpurun_out/r3prof/march_b8.txt $OUT/
bash scripts/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc/summary.json profiles/pmc_traffic.json
cp gpurun_out/pmc/summary.json $OUT/pmc_traffic.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20_warmup5.json 2> $OUT/bench_steps20_warmup5.err
python scripts/run_configs.py 2 3 5 > $OUT/configs_2_3_5.json 2> $OUT/configs.err
python scripts/r3_config5_probe.py 100 > $OUT/config5_probe.txt 2>&1
tail -c 600 $OUT/bench_steps20_warmup5.json
