#!/bin/bash
# round 6, GPU call 54: the rule widened to sources with at most ONE in-edge (siblings of one parent carry equal scores) — its tests, the
# partition / hub / parity tests, hub rows and plan time at scale 26 / 22
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "equal_terms or python_front" 2>&1 | grep -a "fan\|rule\|sibling\|ranks\|passed\|failed\|rror" | cut -c1-230
timeout 1500 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'hub rows', h.get('hub_rows'), 'hub edges', h.get('hub_edges'), 'plan_build_ms', d['config'].get('plan_build_ms'))"; }
timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 26"
timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 22"
timeout 300 python bench.py --scale 24 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 24"
