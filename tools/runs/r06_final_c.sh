#!/bin/bash
# round 6, final records (c): the default line with bench.py's step through gm_pr_sweep (the error folded into the sweep's launches), behind
# the whole suite and smoke() as the driver runs it; the page_rank() drop-in call at scale 26 with both kinds of sweep
OUT=gpurun_out/r06fc; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256; sha256sum bench.py >> $OUT/lib.sha256
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v['roofline'].get('frac'), v['roofline'].get('traffic'), v.get('ms_result_left_on_device')) for k, v in d['extra'].items() if isinstance(v, dict)})"
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --prapi-scale 26 --oracle 0 > $OUT/prapi26.json 2> $OUT/prapi26.err; python -c "
import json; print(json.load(open('$OUT/prapi26.json'))['page_rank_api'])"
