#!/bin/bash
# round 2, GPU call 35: triangle count with the bitmap rows in LDS
OUT=gpurun_out/r02ai; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "triangle or tc or graph_mate or prelude or robust" > $OUT/pytest_tc.log 2>&1; grep -a "passed\|failed" $OUT/pytest_tc.log | tail -2
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run default X=1
run item1024 GM_TC_ITEM=1024
run item16384 GM_TC_ITEM=16384
run k65536 GM_TC_K=65536
run k0 GM_TC_K=0
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --tc-scale 22 --oracle 1 --reps 3 > $OUT/tc22.json 2> $OUT/tc22.err
python -c "import json; d=json.load(open('$OUT/tc22.json'))['tc']; print('scale 22', round(d['ms'],2), 'ms', d['triangles'], d['parity'])"
