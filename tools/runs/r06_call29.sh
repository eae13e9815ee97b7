#!/bin/bash
# round 6, GPU call 29: the dispatches of a block-Gauss-Seidel iteration (the default page_rank() call) at scale 26 and 22, and the
# multi-rank bench path over gloo on one GPU after bench.py's timing change
OUT=gpurun_out/r06ab; mkdir -p $OUT; export TMPDIR=/tmp
for s in 26 22; do
timeout -s KILL 400 rocprofv3 --kernel-trace -d $OUT/g$s -o t -- python tools/gs_call.py $s 2 > $OUT/g$s.log 2>&1; grep -a "^scale" $OUT/g$s.log
python tools/timeline.py $OUT/g$s 9 > $OUT/timeline_gs$s.txt 2>&1; cut -c1-100 $OUT/timeline_gs$s.txt | head -70
done
find $OUT -name "*.db" -delete
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 2> $OUT/gloo2.err | tail -1 | cut -c1-300
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --steps 20 --warmup 5 2> $OUT/gloo8.err | tail -1 | cut -c1-300
