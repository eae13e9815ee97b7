#!/bin/bash
# round 4, GPU call 25: accumulate kernel without a branch per entry (padding to one more accumulator) against the branchy form:
# parity tests, then both forms in alternating fresh processes + kernel stats
OUT=gpurun_out/r04w; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hub_order.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2; grep -a -B30 "Error\|assert " $OUT/pytest.txt | head -40 | cut -c1-200
for rep in 1 2 3; do for br in 0 1; do
  GM_PB_ACC_BRANCHY=$br timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('branchy $br:', d['ms_per_step'], d['roofline']['frac'], c['value_stream_placement'].get('draw_best_us'), c['value_stream_placement'].get('level'))"
done; done
for br in 0 1; do
  GM_PB_ACC_BRANCHY=$br timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace$br -o trace -- python bench.py --cpu-sweeps 0 --algos 0 > $OUT/trace$br.log 2>&1
  DB=$(find $OUT/trace$br -name "*.db" | head -1); echo "== branchy $br"; [ -n "$DB" ] && python tools/rocpd_summary.py $DB 8 | cut -c1-150 | grep "gm::pb_[abh]"
done
find $OUT -name "*.db" -delete
