#!/bin/bash
# round 4, GPU call 33: which rows are long (GM_PB_HUB_LONG) and how many terms a hub group holds (GM_PB_HUB_GROUP), re-measured
# with the 96-VGPR hub kernels that share a CU with the accumulate kernel
OUT=gpurun_out/r04zb; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config']['hub_rows_in_reference_order']; print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'groups', h['hub_groups'], 'long', h['long_rows'], h['long_row_terms'])"; }
for sc in 26 22; do
for cfg in "X=1" "GM_PB_HUB_LONG=16384" "GM_PB_HUB_LONG=8192" "GM_PB_HUB_GROUP=131072" "GM_PB_HUB_GROUP=65536" "GM_PB_HUB_LONG=8192 GM_PB_HUB_GROUP=131072" "X=1"; do
  env $cfg timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc $cfg"
done
done
for cfg in "GM_PB_HUB_LONG=8192" ; do
  env $cfg GM_PB_ANYORDER=1 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale 26 --steps 10 > $OUT/trace.log 2>&1
  echo "serial timeline scale 26 $cfg"; python tools/timeline.py $OUT/trace 1; rm -rf $OUT/trace
done
