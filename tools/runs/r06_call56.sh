#!/bin/bash
# round 6, GPU call 56: stress of the rule's plumbing — the GPU suite's partition / hub / parity files with the rule's threshold at 64
# (many RMAT rows flagged at the small test scales): which comparisons between partitioned runs and the single engine still hold
export TMPDIR=/tmp
GM_PB_HUB_LEAVES=64 timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_hub_order.py tests/test_gpu_hub_adversarial.py -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|rror" | tail -14 | cut -c1-200
