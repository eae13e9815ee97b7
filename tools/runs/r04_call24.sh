#!/bin/bash
# round 4, GPU call 24: hub groups' hot terms off the value stream, A/B with kernel stats (alternating fresh processes on one box)
OUT=gpurun_out/r04v; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for hh in 1 0; do
  GM_PB_HUB_HOT=$hh timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('hub hot $hh:', d['ms_per_step'], d['roofline']['frac'], c['value_stream_placement'].get('draw_best_us'), c['value_stream_placement'].get('level'), c['plan_build_ms'], c['plan_rebuild_ms'])"
done; done
for hh in 1 0; do
  GM_PB_HUB_HOT=$hh timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace$hh -o trace -- python bench.py --cpu-sweeps 0 --algos 0 > $OUT/trace$hh.log 2>&1
  DB=$(find $OUT/trace$hh -name "*.db" | head -1); echo "== hub hot $hh"; [ -n "$DB" ] && python tools/rocpd_summary.py $DB 8 | cut -c1-150 | grep "gm::pb_[abh]"
done
GM_LOG=1 timeout 200 python bench.py --cpu-sweeps 0 --algos 0 --steps 2 2>&1 | grep -a "pb plan" | head -30 | cut -c1-150
find $OUT -name "*.db" -delete
