#!/bin/bash
# round 6, GPU call 12: FETCH_SIZE / WRITE_SIZE / L2 misses against known byte counts for the access patterns of the triangle count, SSSP and
# WCC (tools/fetchcal.hip): random 128-byte records, random 4-byte probes (plain and sc1), 8- and 16-byte-per-lane streams, random stores
OUT=gpurun_out/r06k; mkdir -p $OUT; export TMPDIR=/tmp
tools/fetchcal > $OUT/fetchcal_timings.txt 2>&1; cat $OUT/fetchcal_timings.txt
rocprofv3 -L > $OUT/counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/counters_list.txt && echo -n "$c "; done; }
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum"; do
  cs=$(have $set); [ -z "$cs" ] && { echo "no counter of: $set"; continue; }
  tag=$(echo $cs | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $cs --kernel-trace -d $OUT/cal_$tag -o p -- tools/fetchcal > $OUT/cal_$tag.log 2>&1
done
python tools/fetchcal_report.py $OUT $OUT/fetchcal.json
find $OUT -name "*.db" -delete
# first calls with / without the code objects loaded ahead (GM_WARM)
for wv in 0 1; do
  GM_WARM=$wv GM_SSSP_TIMES=1 timeout 600 python tools/bench_algos.py --skip prapi --oracle 0 --tc-oracle 0 > $OUT/algos_warm$wv.json 2> $OUT/algos_warm$wv.err
  python -c "
import json; d=json.load(open('$OUT/algos_warm$wv.json'))
print('GM_WARM=$wv', {k: {x: round(v, 2) for x, v in d[k].items() if isinstance(v, float) and ('first' in x or x == 'ms' or 'second' in x)} for k in ('wcc', 'sssp', 'tc') if k in d})"
  grep -a "^sssp: init" $OUT/algos_warm$wv.err | head -1
done
