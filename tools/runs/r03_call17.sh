#!/bin/bash
# round 3, GPU call 17: hub kernel with the masks off the hot path and replay for long chains only: bench + kernel stats, debug of the random case
OUT=gpurun_out/r03q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q -s > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed|^E  |random graphs|giant" $OUT/pytest_hub.log | tail -20
for k in 1 2; do timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 > $OUT/bench$k.json; python - <<PY
import json
d=json.load(open('$OUT/bench$k.json')); print('run $k', d['ms_per_step'], d['roofline']['frac'])
PY
done
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 6 > $OUT/kernel_stats.txt; cat $OUT/kernel_stats.txt | cut -c1-60,110-170
find $OUT -name "*.db" -size +20M -delete
