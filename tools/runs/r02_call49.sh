#!/bin/bash
# round 2, GPU call 50: triangle count, groups draw their lists from a counter
OUT=gpurun_out/r02aw; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "triangle or tc or graph_mate or prelude or robust or wcc" > $OUT/pytest_tc.log 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest_tc.log | tail -3
run() { name=$1; shift
  env "$@" timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 --reps 3 > $OUT/tc_$name.json 2> $OUT/tc_$name.err
  python -c "import json; d=json.load(open('$OUT/tc_$name.json'))['tc']; print('$name', round(d['ms'],2), 'ms', d['triangles'])"
}
run default X=1
run item4096 GM_TC_ITEM=4096
run item2048 GM_TC_ITEM=2048
run item512 GM_TC_ITEM=512
run b256g8 GM_TC_SHAPE=256,8,4
run b1024g8_item4096 GM_TC_SHAPE=1024,8,4 GM_TC_ITEM=4096
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,sssp,tc --oracle 0 --reps 5 > $OUT/wcc.json 2> $OUT/wcc.err
python -c "import json; d=json.load(open('$OUT/wcc.json'))['wcc']; print('wcc', round(d['ms'],3), 'ms baseline', round(d['baseline_ms'],3), d['afforest_eq_baseline'])"
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --tc-scale 22 --oracle 1 --reps 3 > $OUT/tc22.json 2> $OUT/tc22.err
python -c "import json; d=json.load(open('$OUT/tc22.json'))['tc']; print('scale 22', round(d['ms'],2), 'ms', d['triangles'], d['parity']['bit_exact_vs_oracle'])"
