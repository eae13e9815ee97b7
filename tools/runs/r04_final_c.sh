#!/bin/bash
# round 4, final records (c), after the last library change: whole GPU suite, smoke(), the default bench line, the PMC passes
# (re-stamped with the library's hash), the 2-rank bench path on one GPU (gloo; functional only)
OUT=gpurun_out/r04fc; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile.sh $OUT/prof --algos 0 > $OUT/profile.log 2>&1; head -7 $OUT/prof/kernel_stats.txt | cut -c1-150
python tools/pmc_traffic.py $OUT/prof/pmc_raw.json scale26_gpus1 9932111876 "round 4, tools/runs/r04_final_c.sh" > $OUT/pmc_traffic.txt 2>&1; tail -5 $OUT/pmc_traffic.txt; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact')) for k, v in d['extra'].items() if isinstance(v, dict)})"
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --algos 0 2> $OUT/gloo2.err | tail -1 > $OUT/gloo2.json; python -c "import json; d = json.loads(open('$OUT/gloo2.json').read()); print('2 gloo ranks on one GPU:', d['n_gpus'], d['ms_per_step'], d['config'].get('partition', '')[:80])" || tail -5 $OUT/gloo2.err
find $OUT -name "*.db" -delete
