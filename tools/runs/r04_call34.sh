#!/bin/bash
# round 4, GPU call 34: do the hub kernels' workgroups keep the accumulate kernel's off the CUs?  Fixed, small grids for the hub
# kernels (GM_PB_LONG_WGS / GM_PB_SEQ_WGS: a workgroup loops over rows / groups) against one workgroup per row / group
OUT=gpurun_out/r04zc; mkdir -p $OUT; export TMPDIR=/tmp
GM_PB_LONG_WGS=3 GM_PB_SEQ_WGS=2 timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; }
for sc in 26 22; do
for cfg in "X=1" "GM_PB_LONG_WGS=256 GM_PB_SEQ_WGS=256" "GM_PB_LONG_WGS=128 GM_PB_SEQ_WGS=512" "GM_PB_LONG_WGS=256 GM_PB_SEQ_WGS=512" "GM_PB_LONG_WGS=64 GM_PB_SEQ_WGS=256" "X=1"; do
  env $cfg timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc $cfg"
done
for cfg in "X=1" "GM_PB_LONG_WGS=256 GM_PB_SEQ_WGS=256"; do
  env $cfg timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale $sc --steps 10 > $OUT/trace.log 2>&1
  echo "timeline scale $sc $cfg"; python tools/timeline.py $OUT/trace 1; rm -rf $OUT/trace
done
done
