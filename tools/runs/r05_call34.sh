#!/bin/bash
# round 5, GPU call 34: 8 gloo ranks, scale 22, part k on its own stream: the first sweep whose scores differ from the single engine's,
# which rows; the same without the hub fork, and with a device synchronisation + barrier after every sweep
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
run() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $2 --streams $3 --sync $4 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('world', 'streams', 'sync', 'rows_that_differ', 'first_sweep_whose_error_differs', 'env')}); print('   first bad:', d['first_sweep_whose_scores_differ'])"; }
run 8 22 1 0
GM_PB_HUB_FORK=0 run 8 22 1 0
run 8 22 1 1
run 8 22 0 0
tail -3 $OUT/debug.err | cut -c1-300
