#!/bin/bash
# round 5, final check (e): the tree as committed — whole GPU suite, smoke(), the default bench line (both counter records in place)
OUT=gpurun_out/r05fe; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v['roofline'].get('frac'), v.get('ms_result_left_on_device')) for k, v in d['extra'].items() if isinstance(v, dict)})"
