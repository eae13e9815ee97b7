#!/bin/bash
# round 6, GPU call 2: the whole GPU suite on the round's first kernel changes (error folded into the sweep's launches, block-Gauss-Seidel
# default call, WCC sample / mode kernels, SSSP lists from the arena), then the default bench line, scale 22 / 24, and the fold A/B
OUT=gpurun_out/r06b; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2700 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -5; grep -a "default config\|block-GS\|to 1e-10" $OUT/pytest.txt | tail
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), d['config'].get('parity', {}).get('max_rel_vs_reference'), {k: (v.get('ms'), v.get('bit_exact'), v.get('ms_result_left_on_device'), v.get('second_call_ms_builds_the_ordered_lists')) for k, v in d.get('extra', {}).items() if isinstance(v, dict)})"; }
( time GM_SSSP_TIMES=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -1 $OUT/bench.json | line default; grep -a "^sssp:" $OUT/bench.err | head -12
for sc in 22 24; do timeout 300 python bench.py --scale $sc --cpu-sweeps 0 --algos 0 > $OUT/bench_$sc.json 2>> $OUT/bench.err; tail -1 $OUT/bench_$sc.json | line "scale $sc"; done
for f in 0 1 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 fold=$f"; done
for f in 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 fold=$f"; done
