#!/bin/bash
# round 6, GPU call 33: pb_hubseq_kernel<512> for the launches whose walks are the critical path (a part's hub rows, a slice's) — hub /
# partition / parity tests, then A/B through the measurement library (GM_PB_SEQ_WIDE=0: 256 threads everywhere, as before)
OUT=gpurun_out/r06af; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'))"; }
for rep in 1 2 3; do for w in "" 0; do
GM_PB_SEQ_WIDE=$w timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 8 wide=[$w]"
GM_PB_SEQ_WIDE=$w timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 6 2>> $OUT/bench.err | tail -1 | line "rank 6 of 8 wide=[$w]"
GM_PB_SEQ_WIDE=$w timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 4 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 4 wide=[$w]"
done; done
for w in "" 0 "" 0; do for s in 22 26; do GM_PB_SEQ_WIDE=$w timeout 300 python tools/gs_time.py $s 2>> $OUT/gs.err | tail -1 | cut -c1-200; done; done
for s in 22 26; do timeout 300 python bench.py --scale $s --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale $s whole graph"; done
