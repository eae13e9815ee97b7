#!/bin/bash
# round 6, GPU call 44: rows with many sources that have no in-edges become hub rows (GM_PB_HUB_LEAVES) — the new test, the hub / partition /
# parity tests, plan build time and hub rows at scale 26 / 22 (no BASELINE row should be flagged)
OUT=gpurun_out/r06al; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "equal_terms" 2>&1 | grep -a "fan\|rule\|passed\|failed\|rror" | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'hub rows', h.get('hub_rows'), 'hub edges', h.get('hub_edges'), 'plan_build_ms', d['config'].get('plan_build_ms'))"; }
for lv in 512 0 512 0; do
GM_PB_HUB_LEAVES=$lv timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 leaves=$lv"
done
GM_PB_HUB_LEAVES=512 timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 leaves=512"
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --steps 3 --warmup 1 2>&1 | grep -a "hub flags" | cut -c1-120
