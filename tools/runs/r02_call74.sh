#!/bin/bash
# round 2, GPU call 74: long chains walked two / four blocks per step: tests, parity at every BASELINE size, sweep times
OUT=gpurun_out/r02bu; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; grep -a "passed\|failed\|long2\|scale" $OUT/pytest.log | tail -8
for s in 22 24 26; do timeout 900 python tools/parity_pagerank.py --scale $s > $OUT/parity_scale$s.json 2> $OUT/parity$s.err; python -c "
import json; d=json.loads(open('$OUT/parity_scale$s.json').read().strip().splitlines()[-1]); print('parity $s', d['max_rel_vs_reference'], d['rows_over_1e-5'], [(c['in_degree'][0], '%.2e' % c['max_rel']) for c in d['by_in_degree'][-4:]])"; done
for a in "X=1" "GM_PB_HUB_LONG2=1000000000 GM_PB_HUB_LONG4=1000000000"; do
env $a timeout 300 python bench.py --cpu-sweeps 0 --scale 22 --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale 22 [$a]', d['ms_per_step'], d['roofline']['frac'])"
env $a timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rank 3 of 8 [$a]', d['ms_per_step'])"
done
timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale 26', d['ms_per_step'], d['roofline']['frac'])"
