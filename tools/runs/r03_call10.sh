#!/bin/bash
# round 3, GPU call 10: standalone probe — scattered 1 KiB writes over a 4 GiB range mapped from subsets of ONE pool of pieces
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 120 tools/scatterprobe 256 256 > $OUT/scatterprobe_256.txt 2>&1; cat $OUT/scatterprobe_256.txt
