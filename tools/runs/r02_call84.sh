#!/bin/bash
# round 2, GPU call 84: value stream in physically contiguous pages or not, alternating fresh processes on one box
OUT=gpurun_out/r02ce; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3 4; do for e in 0 1; do
GM_PB_VALS_CONTIG=$e timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contig $e run $k', d['ms_per_step'], d['roofline']['frac'])"
done; done
