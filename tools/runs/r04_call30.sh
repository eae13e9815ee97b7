#!/bin/bash
# round 4, GPU call 30: the accumulate phase's fork / join without a second stream (hipExtAnyOrderLaunch, GM_PB_ANYORDER=1)
OUT=gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp
GM_PB_ANYORDER=1 timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q -m gpu -k "not sssp and not wcc and not triangle" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), '')"; }
for sc in 22 26; do
  for rep in 1 2; do for any in 0 1; do
    GM_PB_ANYORDER=$any timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc anyorder $any"
  done; done
  GM_PB_ANYORDER=1 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace$sc -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale $sc --steps 10 > $OUT/trace$sc.log 2>&1
  python tools/timeline.py $OUT/trace$sc 1 > $OUT/timeline$sc.txt 2>&1; cat $OUT/timeline$sc.txt
done
for any in 0 1; do GM_PB_ANYORDER=$any timeout 600 python tools/partition_emulated.py --scale 26 --parts 8 2>&1 >/dev/null | tr '\n' ' '; echo; done
find $OUT -name "*.db" -delete
