#!/bin/bash
# round 2, GPU call 9: branchless hub kernel, one stream
OUT=gpurun_out/r02i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_multi.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; grep -E "scale|passed|failed|Error" $OUT/pytest.log | tail -8
for s in 26 22; do
for hub in 4096 0; do
GM_PB_HUB_DEG=$hub timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${s}_$hub -o trace -- python bench.py --cpu-sweeps 0 --scale $s > $OUT/trace${s}_$hub.log 2>&1
grep '^{' $OUT/trace${s}_$hub.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('scale $s hub_deg $hub ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan_ms', c['plan_build_ms'], 'hot', c['hot_sources'], 'entries', c['value_entries'])"
DB=$(find $OUT/trace${s}_$hub -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 4 > $OUT/kernel_stats${s}_$hub.txt
cat $OUT/kernel_stats${s}_$hub.txt | cut -c1-50,105-160 | tail -4 | grep "pb_"
done
done
for s in 24 26; do timeout 300 python bench.py --cpu-sweeps 0 --scale $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale $s (no profiler) ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan_ms', d['config']['plan_build_ms'])"; done
find $OUT -name "*.db" -size +20M -delete
