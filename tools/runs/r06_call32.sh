#!/bin/bash
# round 6, GPU call 32: pb_hubseq_kernel with a block's ranges loaded once, a round ahead, and the rows' LDS stretches skewed over the
# 16-byte slots — hub / partition tests (bits), then the kernel's average duration in the block-Gauss-Seidel call, in the synchronous
# sweep and in an emulated rank of 8 (measurement library: GM_PB_SEQ_SKEW=0 = the old layout)
OUT=gpurun_out/r06ae; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
export GRAPH_MI355X_LIB=$PWD/graph_amd/libgraph_mi355x_measure.so
kern() { python tools/rocpd_summary.py $1 12 | grep -a -E 'hubseq_kernel|hublong_kernel|pb_accum_kernel|pb_bin_kernel' | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' '; }
for skew in 1 0; do
for s in 22 26; do
GM_PB_SEQ_SKEW=$skew timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python tools/gs_call.py $s 2 > $OUT/g.log 2>&1
echo "skew $skew GS scale $s: $(grep -a 'call 1' $OUT/g.log | cut -c1-60) | $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
GM_PB_SEQ_SKEW=$skew timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python bench.py --scale $s --cpu-sweeps 0 --algos 0 > $OUT/g.log 2>&1
echo "skew $skew sync scale $s: $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
done
GM_PB_SEQ_SKEW=$skew timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 --no-piece-events > $OUT/g.log 2>&1
echo "skew $skew rank 0 of 8: $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
done
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'))"; }
for rep in 1 2; do for skew in 1 0; do
GM_PB_SEQ_SKEW=$skew timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 skew=$skew"
GM_PB_SEQ_SKEW=$skew timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 skew=$skew"
GM_PB_SEQ_SKEW=$skew timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 8 skew=$skew"
done; done
for skew in 1 0; do for s in 22 26; do GM_PB_SEQ_SKEW=$skew timeout 300 python tools/gs_time.py $s 2>> $OUT/gs.err | tail -1 | cut -c1-200; done; done
