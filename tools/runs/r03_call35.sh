#!/bin/bash
# round 3, GPU call 35: long chains on their own stream: tests, whole graph at scale 26 / 22, slices of an 8-way partition, timeline
OUT=gpurun_out/r03zc; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed|^E  " $OUT/pytest_hub.log | tail -8
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'])"; }
for p in 1 0 1 0; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "par $p:"; done
for p in 1 0 1 0; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --scale 22 --cpu-sweeps 0 2>/dev/null | tail -1 | line "scale 22 par $p:"; done
for r in 0 3 4 5; do for p in 0 1; do GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $r 2>/dev/null | tail -1 | line "8 parts rank $r par $p:"; done; done
cd /tmp && GM_PB_HUB_PAR=1 timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/tracef -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 --scale 22 > $GRAFT_REPO_ROOT/$OUT/tracef.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/tracef 1 > $OUT/timeline_scale22_par1.txt 2>&1; echo "== scale 22 par 1"; cat $OUT/timeline_scale22_par1.txt
find $OUT -name "*.db" -size +20M -delete
