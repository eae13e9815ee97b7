#!/bin/bash
# round 4, GPU call 38: SSSP with every list ordered by weight (light edges = a prefix): parity, time against CSR order
OUT=gpurun_out/r04zg; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
for ord in 0 1; do
  GM_SSSP_ORDER=$ord GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp_$ord.json 2> $OUT/sssp_$ord.err
  python -c "import json; d=json.load(open('$OUT/sssp_$ord.json'))['sssp']; print('order $ord:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1))"
  grep "sssp:" $OUT/sssp_$ord.err | tail -3
  GM_SSSP_ORDER=$ord GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2>&1 >/dev/null | grep "sssp:" | tail -1
done
for adapt in "0.3,1.2" "0.2,0.8" "0.5,2" "1.5,6"; do
  GM_SSSP_ADAPT=$(python -c "lo,hi='$adapt'.split(','); print(f'{float(lo)*268.4:.0f},{float(hi)*268.4:.0f}')") timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 2>/dev/null | python -c "import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['sssp']; print('ordered, band $adapt x m:', round(d['ms'],3))"
done
