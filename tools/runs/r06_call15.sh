#!/bin/bash
# round 6, GPU call 15: what a block-Gauss-Seidel call costs and where — scale 22 / 26, hub kernels forked beside each block's accumulate
# kernel or in line, K = 16 / 8 / 4 blocks, hub rows per block or beside block 0
OUT=gpurun_out/r06n; mkdir -p $OUT; export TMPDIR=/tmp
for sc in 22 26; do
  python tools/gs_time.py $sc 2>&1 | grep "^scale"
  GM_PB_HUB_FORK=0 python tools/gs_time.py $sc 2>&1 | grep "^scale"
  GM_PR_BLOCK_GS=8 python tools/gs_time.py $sc 2>&1 | grep "^scale"
  GM_PR_BLOCK_GS=8 GM_PB_HUB_FORK=0 python tools/gs_time.py $sc 2>&1 | grep "^scale"
  GM_PR_BLOCK_GS=4 python tools/gs_time.py $sc 2>&1 | grep "^scale"
  GM_PR_GS_HUBS=0 python tools/gs_time.py $sc 2>&1 | grep "^scale"
done | tee $OUT/gs_time.txt
python tools/gs_time.py 26 200 1e-10 2>&1 | grep "^scale" | tee -a $OUT/gs_time.txt
GM_PB_HUB_FORK=0 python tools/gs_time.py 26 200 1e-10 2>&1 | grep "^scale" | tee -a $OUT/gs_time.txt
