#!/bin/bash
# round 4, GPU call 45: SSSP: the cut set in the middle of the first large phase and binding the short lists as well
OUT=gpurun_out/r04zn; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
for cfg in "GM_SSSP_CUT=8" "GM_SSSP_CUT=4" "GM_SSSP_CUT=16" "GM_SSSP_CUT=32" "GM_SSSP_CUT=8"; do
  env $cfg GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp.json 2> $OUT/sssp.err
  python -c "import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('$cfg:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1))"
  grep "sssp:" $OUT/sssp.err | head -1; grep "sssp:" $OUT/sssp.err | tail -1
done
GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2> $OUT/stats.err > /dev/null
grep "sssp round" $OUT/stats.err | tail -73 | awk '{printf "%s thr=%s %sms %s%s| ", $3, $5, $6, $8, ($12!=""?" "$12$13:"")} END{print ""}' | fold -w 220 | tail -9
grep "sssp:" $OUT/stats.err | tail -1
