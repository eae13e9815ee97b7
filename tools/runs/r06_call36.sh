#!/bin/bash
# round 6, GPU call 36: the walk's LDS reads as inline assembly with hand-placed waits (the next step's reads stay in flight under the
# adds of the step at hand) — hub / partition / parity tests (bits), then pb_hubseq_kernel's average duration in the block-Gauss-Seidel
# call, the synchronous sweep and an emulated rank of 8, and the wall clocks
OUT=gpurun_out/r06ai; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
kern() { python tools/rocpd_summary.py $1 12 | grep -a -E 'hubseq_kernel|hublong_kernel|pb_accum_kernel|pb_bin_kernel' | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' '; }
for s in 22 26; do
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python tools/gs_call.py $s 2 > $OUT/g.log 2>&1
echo "GS scale $s: $(grep -a 'call 1' $OUT/g.log | cut -c1-60) | $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python bench.py --scale $s --cpu-sweeps 0 --algos 0 > $OUT/g.log 2>&1
echo "sync scale $s: $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
done
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 --no-piece-events > $OUT/g.log 2>&1
echo "rank 0 of 8: $(kern $(find $OUT/g -name '*.db' | head -1))"; rm -rf $OUT/g
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'))"; }
for rep in 1 2; do
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26"
timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22"
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 8"
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 6 2>> $OUT/bench.err | tail -1 | line "rank 6 of 8"
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 4 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 4"
done
for s in 22 26; do timeout 300 python tools/gs_time.py $s 2>> $OUT/gs.err | tail -1 | cut -c1-200; done
