#!/bin/bash
# round 2, GPU call 7: what does the hub kernel cost when it runs alone (no second stream)?
OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_cpp_prelude.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for s in 26 22; do
for fork in 1 0; do
GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${s}_$fork -o trace -- python bench.py --cpu-sweeps 0 --scale $s > $OUT/trace${s}_$fork.log 2>&1
grep '^{' $OUT/trace${s}_$fork.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('scale $s fork $fork ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan_ms', c['plan_build_ms'], 'hot', c['hot_sources'], 'entries', c['value_entries'], c['hub_rows_in_reference_order'])"
DB=$(find $OUT/trace${s}_$fork -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 4 > $OUT/kernel_stats${s}_$fork.txt
cat $OUT/kernel_stats${s}_$fork.txt | cut -c1-50,105-160 | tail -4
done
done
find $OUT -name "*.db" -size +20M -delete
