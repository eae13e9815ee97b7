#!/bin/bash
# round 5, GPU call 3: tc_rows_kernel with workgroups that draw their items and keep the bit row
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/tc_ab.py 24 "" "GM_TC_PERSIST=0" "GM_TC_WAVES=1" "GM_TC_WAVES=4" "GM_TC_WAVES=16" > $OUT/tc_ab24.txt 2>&1; grep -a "best of" $OUT/tc_ab24.txt
timeout 600 python tools/tc_ab.py 22 "" "GM_TC_PERSIST=0" > $OUT/tc_ab22.txt 2>&1; grep -a "best of" $OUT/tc_ab22.txt
