#!/bin/bash
# round 2, GPU call 17: key layout with the slot in the unsorted high bits: whole GPU suite + plan time
OUT=gpurun_out/r02q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_all.log 2>&1; tail -4 $OUT/pytest_all.log
for i in 1 2; do GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 --scale 26 2> $OUT/plan$i.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale 26 ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan_ms', d['config']['plan_build_ms'])"; grep "pb plan" $OUT/plan$i.err | head -12; done
