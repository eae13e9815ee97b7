#!/bin/bash
# round 6, GPU call 18: bench.py's multi-rank path (one process per rank, PiecewiseExchange in order) after this round's edits, as 2 and 8 gloo
# ranks on ONE GPU at scale 22 (functional record: its time means nothing), and the driver's own launch line with torch.distributed.run
OUT=gpurun_out/r06q; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --steps 10 --warmup 2 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "2 ranks rc=$?"; tail -1 $OUT/bench_2ranks.json | cut -c1-600
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --backend gloo --single-device 1 --scale 22 --steps 10 --warmup 2 > $OUT/bench_8ranks.json 2> $OUT/bench_8ranks.err; echo "8 ranks rc=$?"; tail -1 $OUT/bench_8ranks.json | cut -c1-600
timeout 300 python bench.py --scale 22 --steps 12 --warmup 0 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('one rank, 12 sweeps: final_sweep_error', d['config'].get('final_sweep_error'))"
