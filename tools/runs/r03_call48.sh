#!/bin/bash
# round 3, GPU call 48: HEAD (candidate search for the value stream, plan temporaries from the arena): whole GPU suite, smoke, six plan
# rebuilds in one process (build times), the default bench line
OUT=gpurun_out/r03head2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python tools/placement11.py 26 6 0 2>&1 | grep -a "^plan"
timeout 900 python bench.py > $OUT/bench_default_scale26.json 2> $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench_default_scale26.json').read().strip().splitlines()[-1]); print('scale 26 ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'], 'cpu', d['cpu_baseline']['value'], 'parity', d['config']['parity']['max_rel_vs_reference'], d['config']['parity']['rows_over_1e-5'], d['config']['value_stream_placement'])"
