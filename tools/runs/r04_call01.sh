#!/bin/bash
# round 4, GPU call 1: (a) the new partitioned-vs-oracle tests (scale 26 x 8 virtual ranks, scale 24 x 2 / 4, scale 18 direct);
# (b) VERDICT r3 item 2's measurement: pb_accum_kernel alone (hub groups in line, GM_PB_HUB_FORK=0) against the co-run case,
# kernel durations and SQ / TCP counters per dispatch
OUT=gpurun_out/r04a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi.py -x -q -s -m gpu \
  -k "scale26 or scale24_partitioned or reference_directly" > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt | cut -c1-250
for fork in 1 0; do
  GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace$fork -o trace -- python bench.py --cpu-sweeps 0 > $OUT/trace$fork.log 2>&1
  tail -1 $OUT/trace$fork.log | cut -c1-400
  DB=$(find $OUT/trace$fork -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 14 > $OUT/kernel_stats_fork$fork.txt; cat $OUT/kernel_stats_fork$fork.txt
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_PENDING_STALL_CYCLES_sum SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_SALU"; do
    i=$((i+1))
    GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc${fork}_$i -o pmc -- python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 > $OUT/pmc${fork}_$i.log 2>&1
    python tools/pmc_by_dispatch.py $OUT/pmc${fork}_$i pb_bin_kernel pb_accum_kernel pb_hub_kernel pb_hubchain > $OUT/pmc${fork}_$i.txt 2>&1; tail -12 $OUT/pmc${fork}_$i.txt | cut -c1-300
    find $OUT/pmc${fork}_$i -name "*.db" -size +20M -delete
  done
done
