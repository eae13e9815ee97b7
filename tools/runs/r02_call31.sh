#!/bin/bash
# round 2, GPU call 31: plan sub-phases
OUT=gpurun_out/r02af; mkdir -p $OUT; export TMPDIR=/tmp
GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 --steps 5 --warmup 2 > $OUT/bench26.json 2> $OUT/bench26.err; grep "pb plan" $OUT/bench26.err
