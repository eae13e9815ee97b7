#!/bin/bash
# round 3, GPU call 31: long chains with one workgroup per block and a ticket walk: bit-identity tests, fall-back counts, PAR=0/1 at scale 26 / 22 and on
# the hub-owning slice of an 8-way partition
OUT=gpurun_out/r03za; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -q > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed|^E  " $OUT/pytest_hub.log | tail -8
timeout 600 python tools/hubpar_debug.py 2>&1 | tail -4
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'])"; }
for p in 1 0 1 0; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "par $p:"; done
for p in 1 0 1 0; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --scale 22 --cpu-sweeps 0 2>/dev/null | tail -1 | line "scale 22 par $p:"; done
for r in 0 3; do for p in 1 0; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $r 2>/dev/null | tail -1 | line "8 parts rank $r par $p:"; done; done
