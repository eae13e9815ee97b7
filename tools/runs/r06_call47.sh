#!/bin/bash
# round 6, GPU call 47: the rule for rows of constant terms ON BY DEFAULT — its test (whole graph, three virtual ranks, rule off), then the
# partition tests
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "equal_terms" 2>&1 | grep -a "fan\|rule\|passed\|failed\|rror" | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hub_order.py tests/test_gpu_hub_adversarial.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
