#!/bin/bash
# round 3, GPU call 61: after the revert: three default-path bench processes must finish (120 s limit each)
for rep in 1 2 3; do timeout 120 python bench.py --cpu-sweeps 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
s=sys.stdin.read()
print('process $rep:', (lambda d: (d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement']['draws_timed']))(json.loads(s)) if s.strip() else 'NO OUTPUT')"; done
