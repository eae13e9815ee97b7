#!/bin/bash
# round 4, final records (a): the whole GPU suite, smoke(), the default bench line (extra included), kernel stats + PMC traffic
# (stamped with the library's hash), the sweep timeline, the scale-22 / 24 lines
OUT=gpurun_out/r04fa; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact')) for k, v in d['extra'].items() if isinstance(v, dict)})"
bash tools/profile.sh $OUT/prof --algos 0 > $OUT/profile.log 2>&1; head -8 $OUT/prof/kernel_stats.txt | cut -c1-150
python tools/pmc_traffic.py $OUT/prof/pmc_raw.json scale26_gpus1 9932111876 "round 4, tools/runs/r04_final_a.sh" > $OUT/pmc_traffic.txt 2>&1; tail -6 $OUT/pmc_traffic.txt; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout 300 python bench.py --cpu-sweeps 0 --algos 0 | tail -1 > $OUT/bench_traffic_check.json; python -c "import json; d = json.loads(open('$OUT/bench_traffic_check.json').read()); print('traffic on the line:', d['roofline']['traffic'], '|', d['roofline']['traffic_source'])"
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/tl -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --steps 10 > $OUT/tl.log 2>&1
python tools/timeline.py $OUT/tl 2 > $OUT/timeline26.txt 2>&1; cat $OUT/timeline26.txt
for sc in 22 24; do timeout 300 python bench.py --scale $sc --algos 0 > $OUT/bench_scale$sc.json 2>/dev/null; tail -1 $OUT/bench_scale$sc.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc:', d['ms_per_step'], d['roofline']['frac'], d['config']['parity']['max_rel_vs_reference'])"; done
find $OUT -name "*.db" -delete
