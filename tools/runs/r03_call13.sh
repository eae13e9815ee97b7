#!/bin/bash
# round 3, GPU call 13: multi (resident + overlapped), hub adversarial tests, then the whole GPU suite and the default bench
OUT=gpurun_out/r03m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hub_adversarial.py -x -q -s > $OUT/pytest_new.log 2>&1; grep -E "passed|failed|Error|error|assert" $OUT/pytest_new.log | tail -15
grep -E "max rel|worst|virtual ranks|Unsorted|seed" $OUT/pytest_new.log | sort | uniq | tail -80 > $OUT/new_tests_numbers.txt; tail -40 $OUT/new_tests_numbers.txt
