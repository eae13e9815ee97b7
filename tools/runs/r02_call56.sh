#!/bin/bash
# round 2, GPU call 56: where a slow first plan build spends its time (three fresh processes)
OUT=gpurun_out/r02bc; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3; do
GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 --steps 3 --warmup 1 > $OUT/b$k.json 2> $OUT/b$k.err
python -c "
import json; d=json.loads(open('$OUT/b$k.json').read().strip().splitlines()[-1]); print('run $k', d['ms_per_step'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"
grep -a "pb plan" $OUT/b$k.err | head -12 | cut -c16-90
done
