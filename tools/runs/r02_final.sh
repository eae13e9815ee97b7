#!/bin/bash
# round 2, final records: the whole GPU suite, smoke, the default bench line (with the CPU leg), scale 22 / 24 lines,
# kernel stats + PMC traffic of the scale-26 sweep, WCC / SSSP / TC with full-size oracle comparisons + kernel stats + PMC,
# the page_rank() drop-in call at scale 26, PageRank parity at scale 26
OUT=gpurun_out/r02final; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default_scale26.json 2> $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench_default_scale26.json').read().strip().splitlines()[-1]); print('scale 26 ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'], 'cpu', d['cpu_baseline']['value'])"
for s in 22 24; do timeout 300 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench_scale$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_scale$s.json').read().strip().splitlines()[-1]); print('scale $s ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"; done
bash tools/profile.sh $OUT/prof26 > $OUT/profile.log 2>&1; tail -14 $OUT/profile.log | cut -c1-60,105-160
python tools/pmc_traffic.py $OUT/prof26/pmc_raw.json scale26_gpus1 9932111876 > $OUT/pmc_traffic_scale26.json 2> $OUT/pmc_traffic.err; cp profiles/pmc_traffic.json $OUT/pmc_traffic_all.json; grep -a "hbm_bytes_per_launch\|traffic_over" $OUT/pmc_traffic_scale26.json
timeout 1200 python tools/bench_algos.py --reps 5 > $OUT/algos.json 2> $OUT/algos.err; python -c "
import json; d=json.load(open('$OUT/algos.json'))
for k in ('wcc','sssp','tc'): print(k, round(d[k]['ms'],3), 'ms frac', d[k]['roofline']['frac'], d[k]['parity']['bit_exact_vs_oracle'])
print(d['page_rank_api'])"
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --prapi-scale 26 > $OUT/prapi26.json 2> $OUT/prapi26.err; cat $OUT/prapi26.json | cut -c1-500
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_algos.py --profile 1 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 40 > $OUT/algos_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python tools/bench_algos.py --profile 1 > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_collect.py $OUT/algos_pmc_raw.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
grep -a "sssp_\|tc_\|wcc_" $OUT/algos_kernel_stats.txt | cut -c1-60,105-160 | head -24
timeout 900 python tools/parity_pagerank.py --scale 26 > $OUT/parity_scale26.json 2> $OUT/parity26.err; python -c "
import json; d=json.loads(open('$OUT/parity_scale26.json').read().strip().splitlines()[-1]); print('parity 26', d['max_rel_vs_reference'], d['rows_over_1e-5'], d['device'], d['reference']['seconds'])"
find $OUT -name "*.db" -size +20M -delete
