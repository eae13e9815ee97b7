#!/bin/bash
# round 6, GPU call 40: the new test (rows of equal terms below the hub threshold: the known deviation pinned)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "equal_terms" 2>&1 | tail -15
