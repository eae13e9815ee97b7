#!/bin/bash
# round 6, GPU call 3: (a) which arena-backed SSSP buffer the memory fault of call 2 follows (GM_SSSP_ARENA masks, the bench's own sequence
# of legs), (b) the block-Gauss-Seidel tests with the hub rows summed per block, (c) fold A/B at scale 22 / 26
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
for mask in 7 6 5 3 0; do
  GM_SSSP_ARENA=$mask GM_SSSP_TIMES=1 timeout 600 python bench.py --cpu-sweeps 0 --tc-oracle 0 > $OUT/bench_mask$mask.json 2> $OUT/bench_mask$mask.err
  echo "mask $mask rc=$? $(grep -ac 'Memory access fault' $OUT/bench_mask$mask.err) faults; $(grep -a '^sssp:' $OUT/bench_mask$mask.err | tr '\n' '|' | cut -c1-400)"
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "scale22 or block_gauss" -s 2>&1 | grep -a "passed\|failed\|default config\|block-GS\|to 1e-10\|Error\|assert" | head -20
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; }
for f in 0 1 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 fold=$f"; done
for f in 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 fold=$f"; done
