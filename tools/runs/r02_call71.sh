#!/bin/bash
# round 2, GPU call 71: the whole GPU suite and smoke() on the final tree
OUT=gpurun_out/r02br; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"
