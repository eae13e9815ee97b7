#!/bin/bash
# round 2, GPU call 13: hub kernel at 64 VGPRs: beside the accumulate kernel (fork) or after it
OUT=gpurun_out/r02m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for s in 26 22; do
for fork in 1 0 1 0; do
GM_PB_HUB_FORK=$fork timeout 300 python bench.py --cpu-sweeps 0 --scale $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale $s fork $fork ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'hot', d['config']['hot_sources'])"
done
GM_PB_HUB_DEG=0 timeout 300 python bench.py --cpu-sweeps 0 --scale $s 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scale $s nohub ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"
for fork in 1 0; do
GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${s}_$fork -o trace -- python bench.py --cpu-sweeps 0 --scale $s > $OUT/trace${s}_$fork.log 2>&1
DB=$(find $OUT/trace${s}_$fork -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 4 > $OUT/kernel_stats${s}_$fork.txt
echo "fork $fork"; cat $OUT/kernel_stats${s}_$fork.txt | cut -c1-50,105-160 | grep "pb_"
done
done
find $OUT -name "*.db" -size +20M -delete
