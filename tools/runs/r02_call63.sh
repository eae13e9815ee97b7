#!/bin/bash
# round 2, GPU call 63: triangle count, lists in flight per group / loads in flight per lane (one box)
OUT=gpurun_out/r02bj; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run u4 GM_TC_SHAPE=512,8,4,4
run u8 GM_TC_SHAPE=512,8,4,8
run m8u4 GM_TC_SHAPE=512,8,8,4
run m8u8 GM_TC_SHAPE=512,8,8,8
run u2 GM_TC_SHAPE=512,8,4,2
run u4b GM_TC_SHAPE=512,8,4,4
