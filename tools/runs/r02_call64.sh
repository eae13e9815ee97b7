#!/bin/bash
# round 2, GPU call 64: triangle count, lists in flight per group / loads in flight per lane (one box)
OUT=gpurun_out/r02bk; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run u2 GM_TC_SHAPE=512,8,4,2
run u1 GM_TC_SHAPE=512,8,4,1
run u3 GM_TC_SHAPE=512,8,4,3
run g16u2 GM_TC_SHAPE=512,16,4,2
run b1024u2 GM_TC_SHAPE=1024,8,4,2
run b256u2 GM_TC_SHAPE=256,8,4,2
run u2b GM_TC_SHAPE=512,8,4,2
