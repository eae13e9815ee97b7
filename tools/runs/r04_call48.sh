#!/bin/bash
# round 4, GPU call 48: SSSP plan on the second call: tests, the three timings at scale 24, kernel trace of a planned call
OUT=gpurun_out/r04zq; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -2 $OUT/pytest_sssp.log
GM_SSSP_TIMES=1 timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 2 --reps 3 > $OUT/sssp.json 2> $OUT/sssp.err
python -c "import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('scale 24:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1), 'second', round(d['second_call_ms_builds_the_ordered_lists'],1), d['parity'], d['roofline']['frac'])"
grep "sssp:" $OUT/sssp.err | head -8
GM_SSSP_ORDER=1 bash tools/runs/r04_call39.sh 2>&1 | tail -2; cp gpurun_out/r04zh/sssp_dispatches.txt $OUT/
