#!/bin/bash
# round 5, GPU call 35: the difference of call 33 (8 gloo ranks, scale 22, part k on its own stream) vanished when every sweep's
# scores were copied (call 34): timing.  Without the copies: how often, and with which of the suspects removed?
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
run() { local w=$1 s=$2; shift 2; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $s "$@" 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('world', 'streams', 'sync', 'gather', 'rows_that_differ', 'first_sweep_whose_error_differs', 'max_rel_score_diff', 'env')})"; }
run 8 22 --streams 1
run 8 22 --streams 1
run 8 22 --streams 1 --gather blocking
run 8 22 --streams 1 --gather main
GM_PB_HUB_FORK=0 run 8 22 --streams 1
run 8 22 --streams 1 --sync 1
run 8 22 --streams 1 --gather blocking
run 8 22 --streams 1 --gather main
tail -3 $OUT/debug.err | cut -c1-300
