#!/bin/bash
# round 5, GPU call 39: one stream per part with 16 (and 8) hardware queues per process: more runs of 8 gloo ranks, scale 22
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
run() { local w=$1 s=$2; shift 2; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $s "$@" 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('world', 'streams', 'rows_that_differ', 'first_sweep_whose_error_differs')}, 'GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES')"; }
export GPU_MAX_HW_QUEUES=16
for i in 1 2 3 4 5 6 7 8; do run 8 22 --streams 1 --sweeps 20; done
export GPU_MAX_HW_QUEUES=8
for i in 1 2 3 4; do run 8 22 --streams 1 --sweeps 20; done
