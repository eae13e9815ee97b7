#!/bin/bash
# round 3, GPU call 4: the value stream spread over 0 ... 200 GiB of physical memory in 256 / 64 MiB pieces
OUT=gpurun_out/r03d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python tools/placement8.py 26 > $OUT/placement8.txt 2>&1; cat $OUT/placement8.txt
