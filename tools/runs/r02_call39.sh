#!/bin/bash
# round 2, GPU call 39: triangle count, shapes of the row kernel and larger K
OUT=gpurun_out/r02al; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run default X=1
run k19 GM_TC_K=524288
run k20 GM_TC_K=1048576
run g8m4 GM_TC_SHAPE=8,4
run g16m2 GM_TC_SHAPE=16,2
run g16m8 GM_TC_SHAPE=16,8
run g32m2 GM_TC_SHAPE=32,2
run g32m4 GM_TC_SHAPE=32,4
run g64m2 GM_TC_SHAPE=64,2
run k19_g32m2 GM_TC_K=524288 GM_TC_SHAPE=32,2
