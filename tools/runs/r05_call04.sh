#!/bin/bash
# round 5, GPU call 4: triangle count, the ranges of tc_rows_kernel on streams of their own
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/tc_ab.py 24 "" "GM_TC_STREAMS=0" "GM_TC_STREAMS=0 GM_TC_PERSIST=0" > $OUT/tc_ab24.txt 2>&1; grep -a "best of" $OUT/tc_ab24.txt
timeout 600 python tools/tc_ab.py 22 "" "GM_TC_STREAMS=0" > $OUT/tc_ab22.txt 2>&1; grep -a "best of" $OUT/tc_ab22.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -x -q -m gpu -k "triangle or tc_ or relabel" > $OUT/pytest_tc.txt 2>&1; tail -2 $OUT/pytest_tc.txt
