#!/bin/bash
# round 6, GPU call 31 (EXPERIMENT library, not the tree's): what a round of pb_hubseq_kernel spends its time on — GM_PB_SEQ_ABLATE bits:
# 1 no walk, 2 no scatter, 4 every load from a cache-hot address, 8 no padding zeros; results wrong by design; the kernel's average
# duration in the block-Gauss-Seidel call (8 launches per iteration, alone most of their time) at scale 22 and 26
OUT=gpurun_out/r06ad; mkdir -p $OUT; export TMPDIR=/tmp
for s in 22 26; do for ab in 0 1 2 4 8 3 6 7 15; do
GM_PB_SEQ_ABLATE=$ab timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/g${s}_$ab -o t -- python tools/gs_call.py $s 2 > $OUT/g.log 2>&1
DB=$(find $OUT/g${s}_$ab -name "*.db" | head -1)
echo "scale $s ablate $ab: $(grep -a 'call 1' $OUT/g.log | cut -c1-60) | $(python tools/rocpd_summary.py $DB 12 | grep -a -E 'hubseq_kernel|hublong_kernel|pb_accum_kernel' | awk '{print $(NF-3), $(NF-1)}' | tr '\n' ' ')"
rm -rf $OUT/g${s}_$ab
done; done
