#!/bin/bash
# round 2, GPU call 60: SSSP with the pending minimum and the bookkeeping in one launch
OUT=gpurun_out/r02bg; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sssp or delta" > $OUT/pytest.log 2>&1; grep -a "passed\|failed" $OUT/pytest.log | tail -2
run() { name=$1; shift
  env "$@" GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 5 > $OUT/sssp_$name.json 2> $OUT/sssp_$name.err
  echo "$name: $(python -c "import json; d=json.load(open('$OUT/sssp_$name.json'))['sssp']; print(round(d['ms'],2))") ms | $(grep 'sssp: setup' $OUT/sssp_$name.err | tail -1)"
}
run settled1 GM_SSSP_SETTLED=1
run settled0 GM_SSSP_SETTLED=0
run settled1b GM_SSSP_SETTLED=1
run settled0b GM_SSSP_SETTLED=0
timeout 600 python tools/bench_algos.py --skip prapi,wcc,tc --reps 3 > $OUT/sssp.json 2> $OUT/sssp.err; python -c "
import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('sssp ms', d['ms'], d['parity'])"
timeout 300 python tools/stress_sssp.py 22 5 2>&1 | tail -1
