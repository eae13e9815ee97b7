#!/bin/bash
# round 5, GPU call 20: thin hub groups (<= 4 rows) summed by the long-row kernel, one item list per row
OUT=gpurun_out/r05r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error\|assert" $OUT/pytest.txt | tail -5
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$1:', d['ms_per_step'], c['hub_rows_in_reference_order'], c['final_sweep_error'])"; }
for t in 4 0 4 0; do for r in 0 1; do GM_PB_HUB_THIN=$t timeout 600 python bench.py --emulate-parts 8 --emulate-rank $r --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "rank $r of 8, thin $t"; done; done
for t in 4 0; do GM_PB_HUB_THIN=$t timeout 600 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 26, thin $t"; done
for t in 4 0 4 0; do GM_PB_HUB_THIN=$t timeout 600 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 22, thin $t"; done
