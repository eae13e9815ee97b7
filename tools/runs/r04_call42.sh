#!/bin/bash
# round 4, GPU call 42: SSSP: heavy rounds up to a cut, the rest in one far round
OUT=gpurun_out/r04zk; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
for cut in 0 8 4 16 32; do
  GM_SSSP_CUT=$cut GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp_$cut.json 2> $OUT/sssp_$cut.err
  python -c "import json; d=json.load(open('$OUT/sssp_$cut.json'))['sssp']; print('cut $cut:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1))"
  grep "sssp:" $OUT/sssp_$cut.err | tail -1
done
GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2> $OUT/stats.err > /dev/null
grep "sssp round" $OUT/stats.err | tail -75 | awk '{printf "%s thr=%s %sms %s%s| ", $3, $5, $6, $8, ($12!=""?" "$12$13:"")} END{print ""}' | fold -w 220 | tail -12
grep "sssp:" $OUT/stats.err | tail -1
