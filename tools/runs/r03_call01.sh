#!/bin/bash
# round 3, GPU call 1: (a) does the memory-side cache keep a freshly written stream?  (b) sweep time per way of
# allocating the value stream (hipMalloc draws, VMM pieces of 2 MiB ... one piece, contiguous), (c) the same engines
# under rocprofv3 --pmc: translation misses, EA stalls, request levels per dispatch
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 tools/mallprobe > $OUT/mallprobe.txt 2>&1; cat $OUT/mallprobe.txt
timeout 600 python tools/placement5.py 26 > $OUT/placement5.txt 2>&1; cat $OUT/placement5.txt
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc$i -o pmc -- python tools/placement5.py 26 --pmc > $OUT/pmc$i.log 2>&1
  tail -3 $OUT/pmc$i.log
  python tools/pmc_by_dispatch.py $OUT/pmc$i > $OUT/pmc$i.txt 2>&1; head -40 $OUT/pmc$i.txt
  find $OUT/pmc$i -name "*.db" -size +20M -delete
done
