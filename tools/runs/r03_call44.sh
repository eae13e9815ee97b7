#!/bin/bash
# round 3, GPU call 44: is a 3.3 ms box a slow box or a placement that fell back?  draws with GM_LOG, arena on / off, on one box
OUT=gpurun_out/r03zh; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['plan_build_ms'])"; }
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/log1.err | tail -1 | line "arena on:"; grep -a "draw\|falling\|arena" $OUT/log1.err | head -12
GM_ARENA=0 timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "arena off:"
timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "arena on:"
GM_PB_HUB_PAR=0 timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "arena on, chains sequential:"
GM_PB_TIERS=1 timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "arena on, one tier:"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -6
