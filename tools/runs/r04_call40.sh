#!/bin/bash
# round 4, GPU call 40: SSSP: light rounds on the weight-ordered lists, the heavy round on the CSR's own
OUT=gpurun_out/r04zi; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -1 $OUT/pytest_sssp.log
for ord in 0 1 0 1; do
  GM_SSSP_ORDER=$ord GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp_$ord.json 2> $OUT/sssp_$ord.err
  python -c "import json; d=json.load(open('$OUT/sssp_$ord.json'))['sssp']; print('order $ord:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1))"
  grep "sssp:" $OUT/sssp_$ord.err | tail -1
done
bash tools/runs/r04_call39.sh 2>&1 | tail -2
