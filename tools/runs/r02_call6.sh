#!/bin/bash
# round 2, GPU call 6: hub kernel with register pre-reduction; multi-GPU entry tests; in-box kernel comparison
OUT=gpurun_out/r02f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_cpp_prelude.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
for s in 26 22; do
for hub in 4096 0; do
GM_PB_HUB_DEG=$hub timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${s}_$hub -o trace -- python bench.py --cpu-sweeps 0 --scale $s > $OUT/trace${s}_$hub.log 2>&1
tail -1 $OUT/trace${s}_$hub.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('scale $s hub_deg $hub ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan_ms', c['plan_build_ms'], 'hot', c['hot_sources'], 'entries', c['value_entries'], c['hub_rows_in_reference_order'])"
DB=$(find $OUT/trace${s}_$hub -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 4 > $OUT/kernel_stats${s}_$hub.txt
cat $OUT/kernel_stats${s}_$hub.txt | cut -c1-50,105-160 | tail -4
done
done
timeout 300 python bench.py --cpu-sweeps 0 --scale 24 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale 24 ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
find $OUT -name "*.db" -size +20M -delete
