#!/bin/bash
# round 6, GPU call 7: the whole GPU suite (SSSP's lists all from the arena, WCC sample kernel back to one node per lane, block-GS tests on
# small graphs), the default line behind it, the scale-28 test's own output, the emulated partition table with the schedule that ships
OUT=gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 3000 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time GM_SSSP_TIMES=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06g/bench.json').read().strip().splitlines()[-1])
    print('default', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), (d['config'].get('parity') or {}).get('max_rel_vs_reference'), d['config'].get('plan_build_ms'))
    for k, v in (d.get('extra') or {}).items():
        if isinstance(v, dict): print('   ', k, v.get('ms'), v.get('best_ms'), v.get('bit_exact'), v.get('ms_result_left_on_device'), v.get('first_call_ms'), v.get('second_call_ms_builds_the_ordered_lists'))
except Exception as e:
    print('bench line unreadable:', e)
PY
grep -a "Memory access fault" $OUT/bench.err | head -2; grep -a "^sssp:" $OUT/bench.err | head -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "scale28" 2>&1 | grep -a "scale 28\|sweep equation\|passed\|failed" | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "block_gauss" 2>&1 | grep -a "block-GS\|passed\|failed" | cut -c1-200
timeout 2400 python tools/partition_emulated.py --scale 26 > $OUT/partition_emulated_scale26.json 2> $OUT/partition.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06g/partition_emulated_scale26.json'))
for t in d['table']:
    print(t['gpus'], 'ranks', [r['sweep_ms'] for r in t['ranks']], 'exchange', t['exchange_ms_model'], 'projected', t['projected_sweep_ms'], t.get('projected_speedup'))
PY
