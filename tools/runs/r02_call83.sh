#!/bin/bash
# round 2, GPU call 83: value stream reserved before or after the plan build, alternating fresh processes on one box
OUT=gpurun_out/r02cd; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3 4 5 6; do for e in 0 1; do
GM_PB_EARLY_VALS=$e timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early $e run $k', d['ms_per_step'], d['roofline']['frac'])"
done; done
