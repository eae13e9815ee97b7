#!/bin/bash
# round 6, GPU call 11: where SSSP's first call spends its 10 ms of set-up; the whole suite with the plan's lists all from the arena
# (allocated up front: GM_SSSP_ARENA default 19); the default line behind the suite
OUT=gpurun_out/r06j; mkdir -p $OUT; export TMPDIR=/tmp
GM_SSSP_TIMES=1 timeout 300 python tools/bench_algos.py --skip wcc,tc,prapi --oracle 2 > $OUT/sssp.json 2> $OUT/sssp.err; grep -a "^sssp:" $OUT/sssp.err | head -8
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 3000 python -m pytest tests -q -m gpu --durations=5 > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time GM_SSSP_TIMES=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06j/bench.json').read().strip().splitlines()[-1])
    print('default', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), (d['config'].get('parity') or {}).get('max_rel_vs_reference'), d['config'].get('plan_build_ms'))
    for k, v in (d.get('extra') or {}).items():
        if isinstance(v, dict): print('   ', k, v.get('ms'), v.get('best_ms'), v.get('bit_exact'), v.get('ms_result_left_on_device'), v.get('first_call_ms'), v.get('second_call_ms_builds_the_ordered_lists'))
except Exception as e:
    print('bench line unreadable:', e)
PY
grep -a "Memory access fault" $OUT/bench.err | head -2
