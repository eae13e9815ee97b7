#!/bin/bash
# round 3, GPU call 16: what the hub kernel's replay / graded passes cost at scale 26: bench with GM_PB_HUB_REPLAY=0/1 and kernel stats
OUT=gpurun_out/r03p; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 0; do
GM_LOG=1 GM_PB_HUB_REPLAY=$r timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/bench_r$r.err | tail -1 > $OUT/bench_r$r.json
python - <<PY
import json
d=json.load(open('$OUT/bench_r$r.json')); c=d['config']
print('replay $r', d['ms_per_step'], d['roofline']['frac'])
PY
grep draw $OUT/bench_r$r.err | head -3
done
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find $OUT/trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 12 > $OUT/kernel_stats.txt; cat $OUT/kernel_stats.txt | cut -c1-60,110-170; tail -1 $OUT/trace.log | cut -c1-200
find $OUT -name "*.db" -size +20M -delete
