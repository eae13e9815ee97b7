#!/bin/bash
# round 2, GPU call 41: the PageRank records of the round — default bench line (with the CPU leg), scale 22 / 24
# lines, kernel stats and PMC traffic of the scale-26 sweep
OUT=gpurun_out/r02an; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default_scale26.json 2> $OUT/bench26.err; tail -c 400 $OUT/bench_default_scale26.json
for s in 22 24; do timeout 300 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench_scale$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_scale$s.json').read().strip().splitlines()[-1]); print('scale $s ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"; done
bash tools/profile.sh $OUT/prof26
python tools/pmc_traffic.py $OUT/prof26/pmc_raw.json scale26_gpus1 9932111876 > $OUT/pmc_traffic_scale26.json 2> $OUT/pmc_traffic.err; cp profiles/pmc_traffic.json $OUT/pmc_traffic_all.json; tail -c 400 $OUT/pmc_traffic_scale26.json
