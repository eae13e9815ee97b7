#!/bin/bash
# round 3, GPU call 7: tiers of hot sources: sweep time and bit-identity of the scores for 1 ... 32 tiers; parity tests
OUT=gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/tiers.py 26 1 2 4 8 13 20 32 > $OUT/tiers26.txt 2>&1; cat $OUT/tiers26.txt
timeout 300 python tools/tiers.py 22 1 2 4 8 16 > $OUT/tiers22.txt 2>&1; cat $OUT/tiers22.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "page_rank" --ignore=tests/test_gpu_fullsize.py > $OUT/pytest_pr.log 2>&1; tail -5 $OUT/pytest_pr.log
