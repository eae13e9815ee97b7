#!/bin/bash
# round 6, GPU call 48: the Python partitioned front hands its slices the source flags — bench.py's multi-rank path over gloo on one GPU
# (2 ranks self-launched, 8 as the driver launches them), an emulated rank, the gloo debug tool against the single engine
OUT=gpurun_out/r06am; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 2> $OUT/gloo2.err | tail -1 | cut -c1-200; tail -2 $OUT/gloo2.err | cut -c1-200
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --steps 20 --warmup 5 2> $OUT/gloo8.err | tail -1 | cut -c1-200
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 2> $OUT/emu.err | tail -1 | cut -c1-160; tail -2 $OUT/emu.err | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_multi.py -q -m gpu -x -k "gloo or processes" 2>&1 | grep -a "passed\|failed\|rror\|deselected" | tail -3
