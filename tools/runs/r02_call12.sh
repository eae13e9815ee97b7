#!/bin/bash
# round 2, GPU call 12: the numbers that get committed: parity at 22/24/26 (default engine), bench lines, kernel
# stats + PMC traffic of the scale-26 sweep, the partitioned paths on one GPU
OUT=gpurun_out/r02l; mkdir -p $OUT; export TMPDIR=/tmp
for s in 22 24 26; do
  timeout 600 python tools/parity_pagerank.py --scale $s --mode auto > $OUT/parity_scale$s.json 2> $OUT/parity$s.err
  python -c "
import json; d=json.load(open('$OUT/parity_scale$s.json')); print('scale $s max_rel', d['max_rel_vs_reference'], 'rows>1e-5', d['rows_over_1e-5'], 'sweeps', d['device']['iterations'], 'first', d['device']['first_call_s'], 'next', d['device']['next_call_s'])"
done
bash tools/profile.sh $OUT/prof26 > $OUT/profile26.log 2>&1; tail -14 $OUT/profile26.log
python tools/pmc_traffic.py $OUT/prof26/pmc_raw.json scale26_gpus1 9932111876 > $OUT/pmc_traffic26.txt 2>&1; tail -5 $OUT/pmc_traffic26.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-400
for s in 22 24; do timeout 300 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench$s.json').read().strip().splitlines()[-1]); print('scale $s ms', d['ms_per_step'], 'frac', d['roofline']['frac'])"; done
timeout 300 python bench.py --cpu-sweeps 0 --emulate-parts 8 > $OUT/bench26_emu8.json 2> $OUT/emu8.err; tail -c 600 $OUT/bench26_emu8.json
timeout 300 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 > $OUT/bench22_gloo2.json 2> $OUT/gloo2.err; tail -c 500 $OUT/bench22_gloo2.json; tail -3 $OUT/gloo2.err
