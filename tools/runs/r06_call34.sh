#!/bin/bash
# round 6, GPU call 34 (measurement library): an emulated rank of 8's two hub kernels each WITHOUT the other (GM_PB_HUB_SKIP=2: no
# long rows, 1: no lane walks; results wrong by design) — are the 160 us of pb_hubseq_kernel its own, or the wait for CU room?
OUT=gpurun_out/r06ag; mkdir -p $OUT; export TMPDIR=/tmp
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
for cfg in "0 ''" "2 ''" "1 ''" "2 0" "0 0"; do set -- $cfg; w=$(eval echo $2)
GM_PB_HUB_SKIP=$1 GM_PB_SEQ_WIDE=$w timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 --no-piece-events > $OUT/t.log 2>&1
echo "== skip=$1 wide=[$w]: $(grep -a ms_per_step $OUT/t.log | python -c "import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"
python tools/timeline.py $OUT/t 1 2>&1 | head -8 | cut -c1-100; rm -rf $OUT/t
done
