#!/bin/bash
# round 4, GPU call 21: the bin kernel's chunk size (entries per workgroup) against its write amplification: sweep time per
# GM_PB_CHUNK, and WRITE_SIZE of the bin kernel for two of them
OUT=gpurun_out/r04t; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for ch in 32768 16384 8192 24576 49152; do
  GM_PB_CHUNK=$ch timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('chunk $ch:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('draw_best_us'), d['config']['value_stream_placement'].get('level'))"
done; done
for ch in 32768 16384; do
  GM_PB_CHUNK=$ch timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --kernel-trace -d $OUT/pmc$ch -o pmc -- python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 --algos 0 > $OUT/pmc$ch.log 2>&1
  python tools/pmc_collect.py $OUT/pmc$ch.json $OUT/pmc$ch > /dev/null; python -c "
import json; d = json.load(open('$OUT/pmc$ch.json'))
for k, v in d.items():
    if 'pb_bin_kernel' in k or 'pb_accum_kernel' in k: print('chunk $ch', k[:40], 'WRITE GB', round(v.get('WRITE_SIZE', 0) * 1024 / 1e9, 3), 'FETCH GB', round(2 * v.get('FETCH_SIZE', 0) * 1024 / 1e9, 3))"
done
find $OUT -name "*.db" -delete
