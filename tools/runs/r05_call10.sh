#!/bin/bash
# round 5, GPU call 10: SSSP plan built by two flat radix sorts (ordered lists + transposition), memory gate, one builder per handle
OUT=gpurun_out/r05j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_robustness.py -x -q -m gpu -k "sssp or delta" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
GM_SSSP_TIMES=1 timeout 600 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 2 > $OUT/sssp.json 2> $OUT/sssp.err; python -c "
import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('sssp scale 24: steady', round(d['ms'],3), 'best', round(d['best_ms'],3), 'first', round(d['first_call_ms'],2), 'second', round(d['second_call_ms_builds_the_ordered_lists'],2), d['parity'])"
grep -a "sssp:" $OUT/sssp.err | head -12
timeout 600 python tools/stress_sssp.py > $OUT/stress.txt 2>&1; tail -3 $OUT/stress.txt
