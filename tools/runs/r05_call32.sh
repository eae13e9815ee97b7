#!/bin/bash
# round 5, GPU call 32: the command line the driver launches for 8 GPUs, at the full scale-26 size, with the 8 ranks on ONE GPU
# (gloo standing in for RCCL): partition-local construction, rank-major exchange, K = 2 regions.  The check: after the same number
# of sweeps the 8 ranks' summed sweep error equals the one rank's to the last digit (every row summed in the same order).
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
( time timeout 900 python bench.py --gpus 8 --backend gloo --single-device 1 --cpu-sweeps 0 --algos 0 --prewarm-ms 0 --steps 20 --warmup 5 2> $OUT/gloo8.err | tail -1 > $OUT/gloo8.json ) 2>&1 | grep real
python -c "import json; d = json.loads(open('$OUT/gloo8.json').read()); c = d['config']; print('8 gloo ranks on one GPU, scale 26:', d['ms_per_step'], c['final_sweep_error'], c.get('construction'), c.get('device_bytes_in_use_peak_per_rank'))" || tail -5 $OUT/gloo8.err
( time timeout 600 python bench.py --cpu-sweeps 0 --algos 0 --prewarm-ms 0 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/one.json ) 2>&1 | grep real
python -c "import json; d = json.loads(open('$OUT/one.json').read()); print('one rank, scale 26, 25 sweeps:', d['ms_per_step'], d['config']['final_sweep_error'])"
