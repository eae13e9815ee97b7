#!/bin/bash
# round 3, GPU call 9: which 1 GiB stretches of physical memory go together (bin kernel over two + two pieces)
OUT=gpurun_out/r03i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/placement10.py 26 160 > $OUT/placement10.txt 2>&1; cat $OUT/placement10.txt
