#!/bin/bash
# round 3, GPU call 5: value stream from a pseudo-random subset of a pool of physical pieces: sizes, pool factors, seeds
OUT=gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python tools/placement9.py 26 > $OUT/placement9.txt 2>&1; cat $OUT/placement9.txt
