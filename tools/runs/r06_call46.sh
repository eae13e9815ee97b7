#!/bin/bash
# round 6, GPU call 46: the whole GPU suite with the rule for rows of constant terms ON (GM_PB_HUB_LEAVES=512): which tests notice
export TMPDIR=/tmp
GM_PB_HUB_LEAVES=512 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -a "passed\|failed\|FAILED\|rror" | tail -12
