#!/bin/bash
# round 3, GPU call 27: one block per step for hub groups of up to 32 rows: parity at scale 18 ... 24 (PB), hub tests, bench
OUT=gpurun_out/r03w; mkdir -p $OUT; export TMPDIR=/tmp
for sc in 18 20 22 24; do
timeout 300 python tools/parity_pagerank.py --scale $sc --mode pb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale $sc: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'], 'worst', [(w['in_degree'], round(w['rel']*1e6,2)) for w in d['worst_rows'][:3]])"
done
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_parity.py -q -k "hub or converged or reference" > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed|^E  " $OUT/pytest_hub.log | tail -5
for k in 1 2; do timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 > $OUT/bench$k.json; python - <<PY
import json
d=json.load(open('$OUT/bench$k.json')); print('run $k', d['ms_per_step'], d['roofline']['frac'])
PY
done
