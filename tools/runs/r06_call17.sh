#!/bin/bash
# round 6, GPU call 17: the triangle count's search kernel BESIDE the row launches (low-priority stream) against behind them: A/B at scale 24,
# and the counts' tests
OUT=gpurun_out/r06p; mkdir -p $OUT; export TMPDIR=/tmp
for o in 0 1 0 1; do GM_TC_OVERLAP=$o timeout 300 python tools/bench_algos.py --skip wcc,sssp,prapi --oracle 0 --reps 5 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read())['tc']; print('GM_TC_OVERLAP=$o', round(d['ms'], 3), round(d['best_ms'], 3), d['triangles'], round(d['first_call_ms'], 2))"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_robustness.py -q -m gpu -k "triangle or tc_ or relabel" 2>&1 | tail -2
