#!/bin/bash
# round 4, GPU call 3: pb_hubseq_kernel with the copy-free walk loop; side streams at the lowest priority (GM_PB_SIDE_PRIO=1, default) against 0
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_hub_adversarial.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt | cut -c1-250
for prio in 1 0; do
  for rep in 1 2; do
    GM_PB_SIDE_PRIO=$prio timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('prio $prio rep $rep:', d['ms_per_step'], d['roofline']['frac'], d['config'].get('value_stream_placement'))"
  done
  GM_PB_SIDE_PRIO=$prio timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace$prio -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace$prio.log 2>&1
  DB=$(find $OUT/trace$prio -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 9 > $OUT/kernel_stats_prio$prio.txt; cut -c1-150 $OUT/kernel_stats_prio$prio.txt | grep "gm::"
done
GM_PB_HUB_FORK=0 timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('hub kernels in line:', d['ms_per_step'], d['roofline']['frac'])"
find $OUT -name "*.db" -size +20M -delete
