#!/bin/bash
# round 6, GPU call 20: is the per-process placement level a property of the STREAM (hardware queue) the sweep runs on?  six processes, each
# timing the same engine's sweeps on the null stream, six pool streams and three high-priority ones
OUT=gpurun_out/r06s; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 300 python tools/stream_level_probe.py 26 2>/dev/null | tee -a $OUT/stream_level.jsonl; done
