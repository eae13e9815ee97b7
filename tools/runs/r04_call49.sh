#!/bin/bash
# round 4, GPU call 49: SSSP pull: sources filtered by their bit in `settled` before their distance is read
OUT=gpurun_out/r04zr; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -1 $OUT/pytest_sssp.log
for f in 0 1 0 1; do
  GM_SSSP_PULL_FILTER=$f GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp.json 2> $OUT/sssp.err
  python -c "import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('filter $f:', round(d['ms'],3), 'ms')"; grep "sssp:" $OUT/sssp.err | tail -1
done
GM_SSSP_ORDER=1 GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2> $OUT/stats.err > /dev/null
grep "far)" -A1 $OUT/stats.err | tail -2
