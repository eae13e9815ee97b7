#!/bin/bash
# round 5, GPU call 14: scale 22 (BASELINE config 2), the bins' size and the hot table against the short sweep's fixed costs
OUT=gpurun_out/r05n; mkdir -p $OUT; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$*:', d['ms_per_step'], d['roofline']['frac'], 'wgs', c['workgroups_per_sweep'], 'hot', c['hot_sources'], c['hot_tiers'], c['hot_edges'], 'values', c['value_entries'])"; }
for rep in 1 2; do
run GM_X=0
run GM_PB_RB=10
run GM_PB_RB=12
run GM_PB_RB=13
run GM_PB_HOT=8192
run GM_PB_HOT=4096
run GM_PB_TIERS=2
run GM_PB_TIERS=3
run GM_PB_HUB_DEG=2048
run GM_PB_CHUNK=16384
done
