#!/bin/bash
# round 4, GPU call 35: which of the hub kernels slows the accumulate kernel beside it, and is it the walker's issue priority?
# (GM_PB_HUB_SKIP gives wrong results by design: 1 = no pb_hubseq_kernel, 2 = no pb_hublong_kernel, 3 = neither)
OUT=gpurun_out/r04zd; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; }
for cfg in "X=1" "GM_PB_HUB_SKIP=3" "GM_PB_HUB_SKIP=1" "GM_PB_HUB_SKIP=2" "GM_PB_SEQ_PRIO=0" "GM_PB_SEQ_PRIO=0 GM_PB_HUB_SKIP=2" "GM_PB_HUB_FORK=0" "X=1"; do
  env $cfg timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --parity 0 --scale 26 2>/dev/null | tail -1 | line "scale 26 $cfg"
done
for cfg in "GM_PB_HUB_SKIP=1" "GM_PB_HUB_SKIP=2" "GM_PB_SEQ_PRIO=0"; do
  env $cfg timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale 26 --steps 10 > $OUT/trace.log 2>&1
  echo "timeline scale 26 $cfg"; python tools/timeline.py $OUT/trace 1 | head -6; rm -rf $OUT/trace
done
