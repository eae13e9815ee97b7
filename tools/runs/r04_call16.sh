#!/bin/bash
# round 4, GPU call 16: gm_page_rank_multi_slices (pieces built without the whole graph), per-rank-local exchange layout, hot gather folded into the bin kernel
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_cpp_prelude.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3; grep -a -B5 "Error\|assert" $OUT/pytest.txt | head -40
for sc in 26 22; do for fold in 1 0; do
  GM_PB_FOLD_HOT=$fold timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc fold $fold:', d['ms_per_step'], d['roofline']['frac'])"
done; done
