#!/bin/bash
# round 6, GPU call 13: call 12's counter passes again (its report lost the two gather kernels to a name match), the databases kept until read
OUT=gpurun_out/r06l; mkdir -p $OUT; export TMPDIR=/tmp
tools/fetchcal > $OUT/fetchcal_timings.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/counters_list.txt && echo -n "$c "; done; }
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  cs=$(have $set); [ -z "$cs" ] && { echo "no counter of: $set"; continue; }
  tag=$(echo $cs | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $cs --kernel-trace -d $OUT/cal_$tag -o p -- tools/fetchcal > $OUT/cal_$tag.log 2>&1
done
python tools/fetchcal_report.py $OUT $OUT/fetchcal.json > $OUT/fetchcal_report.txt; cat $OUT/fetchcal_report.txt
find $OUT -name "*.db" -delete
