#!/bin/bash
# round 3, records 3: WCC / SSSP / TC lines with threaded CPU legs (call final_b died on a shadowed name in tools/bench_algos.py)
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/bench_algos.py --reps 5 > $OUT/algos.json 2> $OUT/algos.err; python -c "
import json; d=json.load(open('$OUT/algos.json'))
for k in ('wcc','sssp','tc'): print(k, round(d[k]['ms'],3), 'ms frac', d[k]['roofline']['frac'], d[k]['parity'], {a:b for a,b in d[k]['cpu_baseline'].items() if a!='sample'})
print(d['page_rank_api'])"
