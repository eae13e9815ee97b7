#!/bin/bash
# round 6, GPU call 21: call 20 again on another box (all six of its processes had the null stream at the fast level) — ten processes, ten pool
# streams each: in a process whose NULL stream is slow, is another stream fast?
OUT=gpurun_out/r06t; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python tools/stream_level_probe.py 26 2>/dev/null | tee -a $OUT/stream_level.jsonl | cut -c1-330; done
