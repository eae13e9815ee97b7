#!/bin/bash
# round 2, GPU call 55: differential fuzz of the whole path after this round's kernel changes
OUT=gpurun_out/r02bb; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python tools/fuzz_parity.py 600 77 3000 20000 > $OUT/fuzz_small.log 2>&1; tail -3 $OUT/fuzz_small.log
timeout 400 python tools/fuzz_parity.py 60 78 60000 600000 > $OUT/fuzz_big.log 2>&1; tail -3 $OUT/fuzz_big.log
GM_TC_K=50 GM_SSSP_COOP=4 GM_SSSP_CHUNK=64 timeout 400 python tools/fuzz_parity.py 300 79 3000 20000 > $OUT/fuzz_knobs.log 2>&1; tail -3 $OUT/fuzz_knobs.log
