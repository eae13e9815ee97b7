#!/bin/bash
# round 2, GPU call 81: several engines (own value stream and vectors) on one plan in one process: does the sweep time depend on where the buffers landed?
OUT=gpurun_out/r02cb; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1; do timeout 300 python tools/placement3.py 2>&1 | tail -5; done
