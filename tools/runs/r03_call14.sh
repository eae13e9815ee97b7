#!/bin/bash
# round 3, GPU call 14: hub kernel with graded first block + replay: adversarial / hub-order / multi tests, then PageRank at scale 22 / 26
OUT=gpurun_out/r03n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q -s > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed" $OUT/pytest_hub.log | tail -3
grep -E "max rel|worst|virtual ranks|Unsorted|seed|emulated|long2" $OUT/pytest_hub.log | sed 's/^\.*//' | sort | uniq > $OUT/hub_numbers.txt; grep -E "giant|seed|Unsorted|worst|scale|long2|virtual" $OUT/hub_numbers.txt
grep -E "^E  |Error" $OUT/pytest_hub.log | head -20
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "page_rank" > $OUT/pytest_full_pr.log 2>&1; grep -E "passed|failed|scale 2" $OUT/pytest_full_pr.log | tail -8
