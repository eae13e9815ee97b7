#!/bin/bash
# round 5, GPU call 31: fuzz with the per-call knobs turned (call 30 also set GM_PB_HUB_DEG=64, which makes hub rows the reference's
# f32 sums while the tool compares with exact row sums: its 89 PageRank "mismatches" of 8e-6 .. 3e-5 are that, see the tool's header)
OUT=gpurun_out/r05y; mkdir -p $OUT; export TMPDIR=/tmp
( time GM_TC_K=50 GM_SSSP_COOP=4 GM_SSSP_CHUNK=64 timeout 200 python tools/fuzz_parity.py 200 503 3000 40000 ) > $OUT/fuzz_d.log 2>&1; tail -5 $OUT/fuzz_d.log
( time timeout 150 python tools/fuzz_parity.py 25 504 30000 300000 ) > $OUT/fuzz_e.log 2>&1; tail -5 $OUT/fuzz_e.log
