#!/bin/bash
# round 2, GPU call 65: triangle count, lists in flight per group / loads in flight per lane (one box)
OUT=gpurun_out/r02bl; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run u1 GM_TC_SHAPE=512,8,4,1
run u4 GM_TC_SHAPE=512,8,4,4
run u1b GM_TC_SHAPE=512,8,4,1
run u4b GM_TC_SHAPE=512,8,4,4
run u1_item4096 GM_TC_SHAPE=512,8,4,1 GM_TC_ITEM=4096
run u1_item1024 GM_TC_SHAPE=512,8,4,1 GM_TC_ITEM=1024
