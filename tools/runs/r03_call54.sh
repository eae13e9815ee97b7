#!/bin/bash
# round 3, GPU call 54: last-resort candidates (fresh 256 MiB VMM pieces, plain hipMalloc) when every arena candidate is slow: fresh processes with the
# draws logged; GM_PB_BW_MIN=9999 forces the whole search in one of them
OUT=gpurun_out/r03zj_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['config']['value_stream_placement']; print('$1', d['ms_per_step'], d['roofline']['frac'], v)"; }
for rep in 1 2 3 4; do GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 --steps 10 --warmup 3 2> $OUT/log$rep.err | tail -1 | line "process $rep:"; grep -a "value stream draw" $OUT/log$rep.err | head -14 | cut -c16-120; done
GM_PB_BW_MIN=9999 GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 --steps 10 --warmup 3 2> $OUT/logf.err | tail -1 | line "forced search:"; grep -a "value stream draw" $OUT/logf.err | head -14 | cut -c16-120
timeout 600 python -m pytest tests/test_gpu_arena.py tests/test_gpu_hub_order.py -q 2>&1 | tail -1
