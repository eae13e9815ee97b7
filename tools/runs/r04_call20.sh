#!/bin/bash
# round 4, GPU call 20: hublong with 16 terms per thread, 96 VGPRs: hub tests, scale 22 / 26 lines and timelines
OUT=gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
for sc in 26 22; do
  for rep in 1 2; do timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; done
  timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace$sc -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale $sc --steps 10 > $OUT/trace$sc.log 2>&1
  python tools/timeline.py $OUT/trace$sc 1 > $OUT/timeline$sc.txt 2>&1; cat $OUT/timeline$sc.txt
done
find $OUT -name "*.db" -delete
