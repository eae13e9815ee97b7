#!/bin/bash
# round 5, GPU call 37: (1) the in-order schedule (every part on the caller's stream), 8 gloo ranks on one GPU, scale 22: how often does
# it differ from the single engine?  (2) part k on its own stream with more hardware queues per process (the runtime maps a process's
# HIP streams onto GPU_MAX_HW_QUEUES = 4 queues; 8 processes x 8 streams share this one GPU)  (3) bench.py --gpus 8 --piece-streams 0, scale 26
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
run() { local w=$1 s=$2; shift 2; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $s "$@" 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('world', 'streams', 'gather', 'rows_that_differ', 'first_sweep_whose_error_differs')}, '$GPU_MAX_HW_QUEUES')"; }
for i in 1 2 3 4 5 6 7 8; do run 8 22 --streams 0 --sweeps 20; done
export GPU_MAX_HW_QUEUES=16
for i in 1 2 3 4 5; do run 8 22 --streams 1 --sweeps 20; done
unset GPU_MAX_HW_QUEUES
timeout 600 python bench.py --gpus 8 --backend gloo --single-device 1 --piece-streams 0 --cpu-sweeps 0 --algos 0 --prewarm-ms 0 --steps 20 --warmup 5 2> $OUT/gloo8s0.err | tail -1 > $OUT/gloo8s0.json
python -c "import json; d = json.loads(open('$OUT/gloo8s0.json').read()); c = d['config']; print('8 gloo ranks on one GPU, scale 26, in-order schedule:', d['ms_per_step'], c['final_sweep_error'], '(one rank: 4.402272355163994e-05)')" || tail -5 $OUT/gloo8s0.err
