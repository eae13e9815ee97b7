#!/bin/bash
# round 2, GPU call 51: triangle count, fixed or drawn assignment of lists to groups, same box
OUT=gpurun_out/r02ax; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run dyn1 GM_TC_DYN=1
run dyn0 GM_TC_DYN=0
run dyn1_again GM_TC_DYN=1
run dyn0_again GM_TC_DYN=0
run dyn1_item2048 GM_TC_DYN=1 GM_TC_ITEM=2048
run dyn0_item2048 GM_TC_DYN=0 GM_TC_ITEM=2048
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,sssp,tc --oracle 0 --reps 5 > $OUT/wcc.json 2> $OUT/wcc.err
python -c "import json; d=json.load(open('$OUT/wcc.json'))['wcc']; print('wcc', round(d['ms'],3), 'ms baseline', round(d['baseline_ms'],3), d['afforest_eq_baseline'])"
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --tc-scale 22 --oracle 1 --reps 3 > $OUT/tc22.json 2> $OUT/tc22.err
python -c "import json; d=json.load(open('$OUT/tc22.json'))['tc']; print('scale 22', round(d['ms'],2), 'ms', d['triangles'], d['parity']['bit_exact_vs_oracle'])"
