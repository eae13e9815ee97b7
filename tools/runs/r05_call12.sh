#!/bin/bash
# round 5, GPU call 12: tc_symmetry_kernel flattened (first call of a graph), the allocation variants behind GM_MEASURE:
# the TC / multi tests, first-call time, the measurement library still builds and ablates
OUT=gpurun_out/r05l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_arena.py -x -q -m gpu -k "triangle or tc_ or relabel or symm or multi or pieces or arena or refused" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 0 > $OUT/tc.json 2> $OUT/tc.err; python -c "
import json; d=json.load(open('$OUT/tc.json'))['tc']; print('tc scale 24: steady', round(d['ms'],3), 'best', round(d['best_ms'],3), 'first call', round(d['first_call_ms'],2), d['triangles'])"
timeout 900 python tools/ablate.py 26 50 60 > $OUT/ablate.txt 2>&1; tail -3 $OUT/ablate.txt
rm -f graph_amd/libgraph_mi355x_measure.so
