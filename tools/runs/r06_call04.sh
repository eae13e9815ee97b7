#!/bin/bash
# round 6, GPU call 4: the memory fault of call 2 (default bench, SSSP's second call) against GM_SSSP_ARENA masks — call 3's runs skipped the
# algorithms' legs (--cpu-sweeps 0) —, then the whole GPU suite and the default line
OUT=gpurun_out/r06d; mkdir -p $OUT; export TMPDIR=/tmp
for mask in 7 0 7 6 5 3; do
  GM_SSSP_ARENA=$mask GM_SSSP_TIMES=1 timeout 600 python bench.py --cpu-sweeps 2 --parity 0 --tc-oracle 0 > $OUT/bench_mask$mask.json 2> $OUT/bench_mask$mask.err
  echo "mask $mask rc=$? $(grep -ac 'Memory access fault' $OUT/bench_mask$mask.err) faults; $(grep -a '^sssp:' $OUT/bench_mask$mask.err | tr '\n' '|' | cut -c1-600)"
done
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2700 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -8; grep -a "scale 28\|sweep equation" $OUT/pytest.txt | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06d/bench.json').read().strip().splitlines()[-1])
    print('default', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), (d['config'].get('parity') or {}).get('max_rel_vs_reference'))
    for k, v in (d.get('extra') or {}).items():
        if isinstance(v, dict): print('   ', k, v.get('ms'), v.get('best_ms'), v.get('bit_exact'), v.get('ms_result_left_on_device'), v.get('first_call_ms'), v.get('second_call_ms_builds_the_ordered_lists'))
except Exception as e:
    print('bench line unreadable:', e)
PY
grep -a "Memory access fault" $OUT/bench.err | head -2
