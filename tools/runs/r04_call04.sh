#!/bin/bash
# round 4, GPU call 4: what slows pb_accum_kernel when the hub kernels run beside it?  Matrix over GM_PB_HUB_SKIP (measurement:
# 1 = no pb_hubseq_kernel, 2 = no long chains, 3 = neither) and GM_PB_HUB_ROOM (LDS left beside an accumulate workgroup)
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$label:', d['ms_per_step'], d['roofline']['frac'], d['config'].get('value_stream_placement', {}).get('draw_best_us'))"
}
run "default" A=1
run "no hub kernels at all" GM_PB_HUB_SKIP=3
run "long chains only" GM_PB_HUB_SKIP=1
run "hubseq only" GM_PB_HUB_SKIP=2
run "room 40 KiB" GM_PB_HUB_ROOM=40960
run "room 60 KiB" GM_PB_HUB_ROOM=61440
run "room 60 KiB, no hub kernels" GM_PB_HUB_ROOM=61440 GM_PB_HUB_SKIP=3
run "default again" A=1
for cfg in "A=1" "GM_PB_HUB_SKIP=2" "GM_PB_HUB_SKIP=1"; do
  env $cfg timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace_$cfg.log 2>&1
  DB=$(find $OUT/trace_$cfg -name "*.db" | head -1); echo "== $cfg"; [ -n "$DB" ] && python tools/rocpd_summary.py $DB 9 | cut -c1-150 | grep "gm::"
done
find $OUT -name "*.db" -size +20M -delete
