#!/bin/bash
# round 3, records 1 of 2: whole GPU suite, smoke, default bench line (CPU leg + in-run parity), scale 22 / 24 lines, kernel stats +
# PMC traffic of the scale-26 sweep
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default_scale26.json 2> $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench_default_scale26.json').read().strip().splitlines()[-1]); print('scale 26 ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'], 'cpu', d['cpu_baseline']['value'], 'parity', d['config']['parity']['max_rel_vs_reference'], d['config']['parity']['rows_over_1e-5'])"
for s in 22 24; do timeout 300 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench_scale$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_scale$s.json').read().strip().splitlines()[-1]); print('scale $s ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"; done
bash tools/profile.sh $OUT/prof26 > $OUT/profile.log 2>&1; tail -16 $OUT/profile.log | cut -c1-60,105-160
python tools/pmc_traffic.py $OUT/prof26/pmc_raw.json scale26_gpus1 9932111876 > $OUT/pmc_traffic_scale26.json 2> $OUT/pmc_traffic.err; cp profiles/pmc_traffic.json $OUT/pmc_traffic_all.json; grep -a "hbm_bytes_per_launch\|traffic_over" $OUT/pmc_traffic_scale26.json
DB=$(find $OUT/prof26/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/timeline.py $DB 1 > $OUT/sweep_timeline_scale26.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
