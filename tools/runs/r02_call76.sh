#!/bin/bash
# round 2, GPU call 76: the same eight processes with torch keeping its cached blocks (no large frees before the plan build)
# (BENCH_KEEP_TORCH_CACHE=1 was a temporary edit of bench.py for this run: skip torch.cuda.empty_cache() after the graph build)
OUT=gpurun_out/r02bw; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
BENCH_KEEP_TORCH_CACHE=1 GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 --steps 3 --warmup 1 > $OUT/b$k.json 2> $OUT/b$k.err
python -c "
import json; d=json.loads(open('$OUT/b$k.json').read().strip().splitlines()[-1]); print('run $k plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"
done
grep -a "pb plan" $OUT/b*.err | awk '{ if ($(NF-1)+0 > 40) print }' | cut -c1-140
