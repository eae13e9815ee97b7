#!/bin/bash
# round 3: one box, what it is and what the sweep does on it: memory-system rates, placement draws, the bench line, sensors
OUT=gpurun_out/r03box_$(date +%H%M%S); mkdir -p $OUT; export TMPDIR=/tmp
hostname > $OUT/box.txt; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1 >> $OUT/box.txt; cat $OUT/box.txt | tr '\n' ' '; echo
./tools/membench 2>/dev/null | grep -a "stream_copy\|bin_like run=  256\|lds_atomic u64 x 16384\|lds_atomic u32\|gather8 uniform table=   1 MiB\|gather8 uniform table=  64" | tail -7
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/log.err > $OUT/bench.json; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'], d['config']['plan_build_ms'])"
grep -a "value stream draw" $OUT/log.err | head -8 | cut -c1-110
rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -a "GPU\[0\]" | grep -ai "junction\|memory\|power\|sclk\|mclk\|fclk" | head -8
