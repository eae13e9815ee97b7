#!/bin/bash
# round 3, GPU call 2: (a) variants of the bin kernel's work order on IDENTICAL pages, four draws; (b) per-channel
# (per TCC instance) request counts and stalls, partial writes, write-backs for six allocations of the value stream
OUT=gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/placement6.py 26 4 > $OUT/placement6.txt 2>&1; cat $OUT/placement6.txt
i=0
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ TCC_REQ" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pmc -- python tools/placement5.py 26 --pmc > $OUT/pmc$i.log 2>&1
  python tools/pmc_by_dispatch.py $OUT/pmc$i pb_bin_kernel pb_accum_kernel > $OUT/pmc$i.txt 2>&1; grep -v "^#" $OUT/pmc$i.txt | head -30
  rm -rf $OUT/pmc$i
done
