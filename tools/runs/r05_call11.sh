#!/bin/bash
# round 5, GPU call 11: whole GPU suite + smoke on the current library; per-call kernel stats + counters of WCC / SSSP / TC
# (-> profiles/algos_traffic.json stamped with this library); the default bench line
OUT=gpurun_out/r05k; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/counters_list.txt && echo -n "$c "; done; }
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/algos_trace -o t -- python tools/bench_algos.py --profile 1 > $OUT/algos_record.json 2> $OUT/algos_trace.err
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  cs=$(have $set); [ -z "$cs" ] && continue
  tag=$(echo $cs | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 500 rocprofv3 --pmc $cs --kernel-trace -d $OUT/algos_pmc_$tag -o p -- python tools/bench_algos.py --profile 1 > $OUT/algos_pmc_$tag.json 2> $OUT/algos_pmc_$tag.err
done
ALGOS_PROFILE_JSON=$OUT/algos_profile.json python tools/algos_profile.py $OUT/algos_record.json $OUT/algos_trace $OUT/algos_pmc_* > $OUT/algos_profile.txt 2>&1
python tools/algos_traffic.py $OUT/algos_profile.json "round 5, tools/runs/r05_call11.sh" > $OUT/algos_traffic.txt 2>&1; cp profiles/algos_traffic.json $OUT/algos_traffic.json
grep -a "^## " $OUT/algos_profile.txt | head -8; grep -a "hbm_bytes_per_call\|kernel_us" $OUT/algos_traffic.txt
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('best_ms'), v.get('bit_exact'), v['roofline'].get('frac'), v['roofline'].get('traffic')) for k, v in d['extra'].items() if isinstance(v, dict)})"
find $OUT -name "*.db" -size +8M -delete
