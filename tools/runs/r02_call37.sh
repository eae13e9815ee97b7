#!/bin/bash
# round 2, GPU call 37: triangle count, cooperative tail
OUT=gpurun_out/r02ak; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "triangle or tc or graph_mate or prelude or robust" > $OUT/pytest_tc.log 2>&1; grep -a "passed\|failed" $OUT/pytest_tc.log | tail -2
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run default X=1
run k131072 GM_TC_K=131072
run k0 GM_TC_K=0
bash tools/runs/r02_call36.sh 2>&1 | tail -8
