#!/bin/bash
# round 2, GPU call 78: does a longer untimed pre-warm change the sweep time a fresh process measures?
OUT=gpurun_out/r02by; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3 4 5; do for pw in 400 3000; do
timeout 300 python bench.py --cpu-sweeps 0 --prewarm-ms $pw 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prewarm $pw run $k', d['ms_per_step'], d['roofline']['frac'])"
done; done
