#!/bin/bash
# round 3, GPU call 8: tiers of hot sources with two tables in LDS: sweep time for 1 ... 48 tiers; exact-row bit-identity; parity tests
OUT=gpurun_out/r03h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/tiers.py 26 1 2 4 8 12 16 24 32 48 > $OUT/tiers26.txt 2>&1; cat $OUT/tiers26.txt
GM_PB_HUB_DEG=0 timeout 600 python tools/tiers.py 24 1 4 16 > $OUT/tiers24_exact.txt 2>&1; cat $OUT/tiers24_exact.txt
timeout 300 python tools/tiers.py 22 1 2 4 8 16 32 > $OUT/tiers22.txt 2>&1; cat $OUT/tiers22.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "page_rank" --ignore=tests/test_gpu_fullsize.py > $OUT/pytest_pr.log 2>&1; grep -E "passed|failed" $OUT/pytest_pr.log
