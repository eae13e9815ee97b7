#!/bin/bash
# round 3, records 2 of 2: WCC / SSSP / TC lines with threaded CPU legs, the emulated partition table, a sweep's kernel timeline
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/bench_algos.py --reps 5 > $OUT/algos.json 2> $OUT/algos.err; python -c "
import json; d=json.load(open('$OUT/algos.json'))
for k in ('wcc','sssp','tc'): print(k, round(d[k]['ms'],3), 'ms frac', d[k]['roofline']['frac'], d[k]['parity']['bit_exact_vs_oracle'], d[k]['cpu_baseline'])
print(d['page_rank_api'])"
timeout 900 python tools/partition_emulated.py > $OUT/partition_emulated_scale26.json 2> $OUT/partition.err; python -c "
import json; d=json.load(open('$OUT/partition_emulated_scale26.json'))
for t in d['table']: print(t['gpus'], t['fastest_rank_ms'], t['slowest_rank_ms'], t['exchange_ms_model'], t['projected_sweep_ms'], t.get('projected_speedup'))"
cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/tl -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 --steps 6 --warmup 2 --prewarm-ms 0 > $GRAFT_REPO_ROOT/$OUT/tl.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/tl 1 > $OUT/sweep_timeline_scale26.txt 2>&1; cat $OUT/sweep_timeline_scale26.txt
find $OUT -name "*.db" -delete
