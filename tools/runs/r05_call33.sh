#!/bin/bash
# round 5, GPU call 33: 8 gloo ranks at scale 26 ended on a sweep error 0.8 % off the single rank's (call 32; 2 ranks at scale 22 had
# matched to the last digit, call 19).  Is it the scores or only the error sum, from which sweep, with part k on its own stream or not?
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
run() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $2 --streams $3 --sync $4 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | cut -c1-700; }
run 4 20 1 0
run 4 20 0 0
run 4 20 1 1
run 2 20 1 0
run 8 22 1 0
run 8 22 0 0
tail -3 $OUT/debug.err | cut -c1-300
