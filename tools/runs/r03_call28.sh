#!/bin/bash
# round 3, GPU call 28: what the graded first block (bit 0) and the one-block steps of few-row groups (bit 1) cost, one box
OUT=gpurun_out/r03x; mkdir -p $OUT; export TMPDIR=/tmp
for g in 3 0 1 3 0; do
GM_PB_HUB_GRADED=$g timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graded $g:', d['ms_per_step'], d['roofline']['frac'])"
done
