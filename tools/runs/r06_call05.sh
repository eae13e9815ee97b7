#!/bin/bash
# round 6, GPU call 5: the whole GPU suite (SSSP's transposed lists back on hipMalloc, in-bounds by scan, per-block hub rows in the
# block-Gauss-Seidel sweeps, scale-28 test), smoke, the default line BEHIND the suite (as the driver runs it), fork-by-stop-event A/B
OUT=gpurun_out/r06e; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 3000 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -8; grep -a "scale 28\|sweep equation\|default config\|block-GS" $OUT/pytest.txt | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time GM_SSSP_TIMES=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06e/bench.json').read().strip().splitlines()[-1])
    print('default', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), (d['config'].get('parity') or {}).get('max_rel_vs_reference'), d['config'].get('plan_build_ms'))
    for k, v in (d.get('extra') or {}).items():
        if isinstance(v, dict): print('   ', k, v.get('ms'), v.get('best_ms'), v.get('bit_exact'), v.get('ms_result_left_on_device'), v.get('first_call_ms'), v.get('second_call_ms_builds_the_ordered_lists'))
except Exception as e:
    print('bench line unreadable:', e)
PY
grep -a "Memory access fault" $OUT/bench.err | head -2; grep -a "^sssp:" $OUT/bench.err | head -4
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; }
for f in 0 1 0 1 0 1; do GM_PB_FORK_STOP=$f timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench2.err | tail -1 | line "scale 22 fork_stop=$f"; done
for f in 0 1 0 1; do GM_PB_FORK_STOP=$f timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench2.err | tail -1 | line "scale 26 fork_stop=$f"; done
GM_PB_FORK_STOP=1 timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -q -m gpu 2>&1 | tail -2
