#!/bin/bash
# round 2, GPU call 66: triangle count, lists in flight per group / loads in flight per lane (one box)
OUT=gpurun_out/r02bm; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run u1 GM_TC_SHAPE=512,8,4,1
run b256 GM_TC_SHAPE=256,8,4,1
run b1024 GM_TC_SHAPE=1024,8,4,1
run g16 GM_TC_SHAPE=512,16,4,1
run m8 GM_TC_SHAPE=512,8,8,1
run k18 GM_TC_K=262144
run u1b GM_TC_SHAPE=512,8,4,1
