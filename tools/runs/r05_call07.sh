#!/bin/bash
# round 5, GPU call 7: where a SHORT sweep's time goes now — dispatch timelines at scale 22 and of an emulated rank of 8 at scale 26
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t22 -o t -- python bench.py --scale 22 --cpu-sweeps 0 --algos 0 > $OUT/t22.log 2>&1
tail -1 $OUT/t22.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale 22 (traced):', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
python tools/timeline.py $OUT/t22 2 | cut -c1-120
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t8 -o t -- python bench.py --emulate-parts 8 --emulate-rank 1 --cpu-sweeps 0 --algos 0 > $OUT/t8.log 2>&1
tail -1 $OUT/t8.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('rank 1 of 8 (traced):', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
python tools/timeline.py $OUT/t8 2 | cut -c1-140
for i in 1 2; do timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale 22:', d['ms_per_step'], d['roofline']['frac'])"; done
find $OUT -name "*.db" -size +8M -delete
