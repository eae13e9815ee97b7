#!/bin/bash
# round 5, GPU call 40: the new guard test of front (b) as 8 processes on one GPU (in-order schedule)
export TMPDIR=/tmp
timeout 110 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "one_process_per_rank" 2>&1 | tail -3
