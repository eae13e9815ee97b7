#!/bin/bash
# round 4, GPU call 58: 2-byte hot records: the accumulate kernel's duration under the tracer (the bin kernel carries the
# placement level of the process, the accumulate kernel does not), alternating processes
OUT=gpurun_out/r04zzz; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for h in 0 1; do
  GM_PB_HOT16=$h timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python bench.py --cpu-sweeps 0 --algos 0 > $OUT/t.log 2>&1
  DB=$(find $OUT/t -name "*.db" | head -1); python tools/rocpd_summary.py $DB 8 2>/dev/null | grep "pb_accum\|pb_bin_kernel\|pb_hubseq\|pb_hublong" | awk -v h=$h '{printf "hot16 %s: %s %s us | ", h, substr($2,1,24), $(NF-1)} END{print ""}'
  tail -1 $OUT/t.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('   line:', d['ms_per_step'], d['config']['value_stream_placement'].get('level'))"
  rm -rf $OUT/t
done; done
