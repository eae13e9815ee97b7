#!/bin/bash
# round 6, GPU call 45: the rule for rows of constant terms ON (GM_PB_HUB_LEAVES=512) in the C++ partitioned front as well (it hands its
# slices the per-slot flags): the partition tests that compare with the single engine bit for bit, and the rest of the hub / parity tests
export TMPDIR=/tmp
GM_PB_HUB_LEAVES=512 timeout 1500 python -m pytest tests/test_gpu_multi.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror\|assert" | tail -6
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hub_order.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
