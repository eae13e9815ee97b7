#!/bin/bash
# round 4, GPU call 52: SSSP pull: the offsets of a lane's 16 nodes read ahead (6.61 against 6.40 ms: reverted)
OUT=gpurun_out/r04zu; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -1 $OUT/pytest_sssp.log
for rep in 1 2; do
  GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp.json 2> $OUT/sssp.err
  python -c "import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('scale 24:', round(d['ms'],3), 'ms')"; grep "sssp:" $OUT/sssp.err | tail -1
done
GM_SSSP_ORDER=1 GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2> $OUT/stats.err > /dev/null
grep "far)" -A1 $OUT/stats.err | tail -2
