#!/bin/bash
# round 3, GPU call 3: sweep time against the position of the value stream inside one 20 GB allocation (256 MiB steps)
OUT=gpurun_out/r03c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/placement7.py 26 16 256 2 > $OUT/placement7.txt 2>&1; cat $OUT/placement7.txt
