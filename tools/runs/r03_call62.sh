#!/bin/bash
# round 3, GPU call 62: the default bench line on the final binaries (record)
OUT=gpurun_out/r03last; mkdir -p $OUT
timeout 200 python bench.py > $OUT/bench_default_scale26.json 2> $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench_default_scale26.json').read().strip().splitlines()[-1]); print('scale 26 ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'], 'cpu', d['cpu_baseline']['value'], 'parity', d['config']['parity']['max_rel_vs_reference'], d['config']['parity']['rows_over_1e-5'], d['config']['value_stream_placement'])"
