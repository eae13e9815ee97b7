#!/bin/bash
# round 4, GPU call 6: hub rows summed the reference's way by both kernels (pb_hubseq_kernel + pb_hublong_kernel): hub tests, bench, kernel trace
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_hub_adversarial.py tests/test_gpu_multi.py -x -q -m gpu -s > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error\|error\|assert" $OUT/pytest.txt | head -20 | cut -c1-250; grep -a "random graphs\|scale 20: one sweep" $OUT/pytest.txt | cut -c1-250
run() { local label=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$label:', d['ms_per_step'], d['roofline']['frac'], c.get('value_stream_placement', {}).get('draw_best_us'), c.get('hub_rows_in_reference_order'))"
}
run "default" A=1
run "default" A=1
run "scale 22" A=1 -- 2>/dev/null
timeout 300 python bench.py --cpu-sweeps 0 --scale 22 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('scale 22:', d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --cpu-sweeps 0 --scale 24 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('scale 24:', d['ms_per_step'], d['roofline']['frac'])"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 9 | cut -c1-150 | grep "gm::"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace22 -o trace -- python bench.py --cpu-sweeps 0 --scale 22 > $OUT/trace22.log 2>&1
DB=$(find $OUT/trace22 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 12 | cut -c1-150 | grep "gm::"
find $OUT -name "*.db" -size +20M -delete
