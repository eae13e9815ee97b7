#!/bin/bash
# round 2, GPU call 61: SSSP schedule bands with the current kernels
OUT=gpurun_out/r02bh; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 5 > $OUT/sssp_$name.json 2> $OUT/sssp_$name.err
  echo "$name: $(python -c "import json; d=json.load(open('$OUT/sssp_$name.json'))['sssp']; print(round(d['ms'],2))") ms | $(grep 'sssp: setup' $OUT/sssp_$name.err | tail -1 | cut -c45-100)"
}
run default X=1
run a100_400 GM_SSSP_ADAPT=100,400
run a200_800 GM_SSSP_ADAPT=200,800
run a300_1200 GM_SSSP_ADAPT=300,1200
run a400_1600 GM_SSSP_ADAPT=400,1600
run a150_2000 GM_SSSP_ADAPT=150,2000
run w16 GM_SSSP_WIDTH=0.0625
run w64 GM_SSSP_WIDTH=0.015625
run w16_a200_800 GM_SSSP_WIDTH=0.0625 GM_SSSP_ADAPT=200,800
run default_again X=1
