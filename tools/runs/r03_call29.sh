#!/bin/bash
# round 3, GPU call 29: long chains walked in parallel: bit-identity tests, hub / multi tests, bench on one box with PAR=0/1, kernel stats
OUT=gpurun_out/r03y; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed|^E  " $OUT/pytest_hub.log | tail -8
for p in 1 0 1 0; do
GM_PB_HUB_PAR=$p timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('par $p:', d['ms_per_step'], d['roofline']['frac'])"
done
GM_PB_HUB_PAR=1 timeout 300 python bench.py --scale 22 --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scale 22 par 1:', d['ms_per_step'], d['roofline']['frac'])"
GM_PB_HUB_PAR=0 timeout 300 python bench.py --scale 22 --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scale 22 par 0:', d['ms_per_step'], d['roofline']['frac'])"
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 14 > $OUT/kernel_stats.txt; cat $OUT/kernel_stats.txt | cut -c1-64,110-170
find $OUT -name "*.db" -size +20M -delete
