#!/bin/bash
# round 5, GPU call 15: one hot table trimmed to the sources that pay for their staging (count >= bins / 2): scales 20 / 22, a rank of 8
OUT=gpurun_out/r05o; mkdir -p $OUT; export TMPDIR=/tmp
run() { sc=$1; shift; env "$@" timeout 300 python bench.py --scale $sc --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('scale $sc $*:', d['ms_per_step'], d['roofline']['frac'], 'hot', c['hot_sources'], c['hot_tiers'], c['hot_edges'], 'values', c['value_entries'], c['final_sweep_error'])"; }
for rep in 1 2 3; do
run 22 GM_PB_HOT_TRIM=1
run 22 GM_PB_HOT_TRIM=0
done
for rep in 1 2; do
run 20 GM_PB_HOT_TRIM=1
run 20 GM_PB_HOT_TRIM=0
run 23 GM_PB_HOT_TRIM=1
run 23 GM_PB_HOT_TRIM=0
done
for t in 1 0; do GM_PB_HOT_TRIM=$t timeout 600 python bench.py --emulate-parts 8 --emulate-rank 1 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('rank 1 of 8, trim $t:', d['ms_per_step'], 'hot', c['hot_sources'], c['hot_tiers'], c['hot_edges'])"; done
