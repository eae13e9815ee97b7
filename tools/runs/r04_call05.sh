#!/bin/bash
# round 4, GPU call 5: pb_hubseq_kernel with two blocks of lookahead; two accumulate workgroups per CU (8192-row bins) with 16384- / 32768-source tiles
OUT=gpurun_out/r04e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_hub_adversarial.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt | cut -c1-250
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$label:', d['ms_per_step'], d['roofline']['frac'], c.get('value_stream_placement', {}).get('draw_best_us'), {k: c[k] for k in c if 'plan' in k or 'hot' in k or 'tier' in k or 'bins' in k})"
}
run "default" A=1
run "hubseq only" GM_PB_HUB_SKIP=2
run "no hub kernels" GM_PB_HUB_SKIP=3
run "8192-row bins" GM_PB_RB=13
run "8192-row bins, 32768-source tiles" GM_PB_RB=13 GM_PB_SLOG=15
run "8192-row bins, 32768-source tiles, 1 wg" GM_PB_RB=13 GM_PB_SLOG=15 GM_PB_WGS=1
run "16384-row bins, 32768-source tiles" GM_PB_SLOG=15
run "default again" A=1
for cfg in "A=1" "GM_PB_RB=13"; do
  env $cfg timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace_$cfg.log 2>&1
  DB=$(find $OUT/trace_$cfg -name "*.db" | head -1); echo "== $cfg"; [ -n "$DB" ] && python tools/rocpd_summary.py $DB 9 | cut -c1-150 | grep "gm::"
done
find $OUT -name "*.db" -size +20M -delete
