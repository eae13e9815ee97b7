#!/bin/bash
# round 2, GPU call 70: triangle count, lists in flight per group / loads in flight per lane (one box)
OUT=gpurun_out/r02bq; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run base X=1
run hub16k_4 GM_TC_HUB=16384,4
run hub16k_2 GM_TC_HUB=16384,2
run hub16k_8 GM_TC_HUB=16384,8
run hub64k_4 GM_TC_HUB=65536,4
run hub64k_2 GM_TC_HUB=65536,2
run base_b X=1
