#!/bin/bash
# round 6, final records (b): WCC / SSSP / TC per-call kernel stats + counters (-> profiles/algos_traffic.json), scale 22 / 24 lines, the
# two timelines, the default line once more with both counter records in place
OUT=gpurun_out/r06fb; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
rocprofv3 -L > $OUT/counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/counters_list.txt && echo -n "$c "; done; }
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/algos_trace -o t -- python tools/bench_algos.py --profile 1 > $OUT/algos_record.json 2> $OUT/algos_trace.err
DB=$(find $OUT/algos_trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 40 > $OUT/algos_kernel_stats.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  cs=$(have $set); [ -z "$cs" ] && continue
  tag=$(echo $cs | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 500 rocprofv3 --pmc $cs --kernel-trace -d $OUT/algos_pmc_$tag -o p -- python tools/bench_algos.py --profile 1 > $OUT/algos_pmc_$tag.json 2> $OUT/algos_pmc_$tag.err
done
ALGOS_PROFILE_JSON=$OUT/algos_profile.json python tools/algos_profile.py $OUT/algos_record.json $OUT/algos_trace $OUT/algos_pmc_* > $OUT/algos_profile.txt 2>&1
python tools/algos_traffic.py $OUT/algos_profile.json "round 6, tools/runs/r06_final_b.sh" > $OUT/algos_traffic.txt 2>&1; cp profiles/algos_traffic.json $OUT/algos_traffic.json
grep -a "^## " $OUT/algos_profile.txt | head -8; grep -a "hbm_bytes_per_call" $OUT/algos_traffic.txt
timeout 1200 python tools/bench_algos.py --reps 5 --tc-oracle 0 > $OUT/algos.json 2> $OUT/algos.err; python -c "
import json; d=json.load(open('$OUT/algos.json'))
for k in ('wcc','sssp','tc'): print(k, round(d[k]['ms'],3), 'best', round(d[k]['best_ms'],3), 'frac', d[k]['roofline']['frac'], 'traffic', d[k]['roofline'].get('traffic'), d[k]['parity']['bit_exact_vs_oracle'], {x: round(v, 2) for x, v in d[k].items() if isinstance(v, float) and ('call_ms' in x or 'device' in x)})
print(d.get('page_rank_api'))"
for s in 22 24; do timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $s > $OUT/bench_scale$s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/bench_scale$s.json').read().strip().splitlines()[-1]); print('scale $s ms', d['ms_per_step'], 'GTEPS', d['value'], 'frac', d['roofline']['frac'])"; done
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t26 -o t -- python bench.py --cpu-sweeps 0 --algos 0 > $OUT/t26.log 2>&1; python tools/timeline.py $OUT/t26 2 > $OUT/timeline26.txt 2>&1; cat $OUT/timeline26.txt | cut -c1-100
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t22 -o t -- python bench.py --scale 22 --cpu-sweeps 0 --algos 0 > $OUT/t22.log 2>&1; python tools/timeline.py $OUT/t22 2 > $OUT/timeline22.txt 2>&1; cat $OUT/timeline22.txt | cut -c1-100
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v['roofline'].get('frac'), v['roofline'].get('traffic'), v.get('ms_result_left_on_device')) for k, v in d['extra'].items() if isinstance(v, dict)})"
find $OUT -name "*.db" -delete
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --prapi-scale 26 --oracle 0 > $OUT/prapi26.json 2> $OUT/prapi26.err; python -c "
import json; print(json.load(open('$OUT/prapi26.json'))['page_rank_api'])"
