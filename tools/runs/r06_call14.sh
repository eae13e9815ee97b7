#!/bin/bash
# round 6, GPU call 14: the error fold measured where the bench can see it (bench.py's step now goes through gm_pr_sweep at N = 1; until
# this call it ran sweep_tiles + sweep_fixup, so call 3's A/B compared nothing) — scale 22 and 26, alternating, and a timeline of each
OUT=gpurun_out/r06m; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['value_stream_placement'].get('level'))"; }
for f in 0 1 0 1 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 fold=$f"; done
for f in 0 1 0 1; do GM_PB_FOLD_ERR=$f timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 26 fold=$f"; done
for f in 0 1 0 1; do GM_PB_FOLD_ERR=1 GM_PB_FORK_STOP=$f timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>> $OUT/bench.err | tail -1 | line "scale 22 fold=1 fork_stop=$f"; done
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t22 -o t -- python bench.py --scale 22 --cpu-sweeps 0 --algos 0 > $OUT/t22.log 2>&1; python tools/timeline.py $OUT/t22 2 > $OUT/timeline22.txt 2>&1; cat $OUT/timeline22.txt | cut -c1-100
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t26 -o t -- python bench.py --cpu-sweeps 0 --algos 0 > $OUT/t26.log 2>&1; python tools/timeline.py $OUT/t26 2 > $OUT/timeline26.txt 2>&1; cat $OUT/timeline26.txt | cut -c1-100
find $OUT -name "*.db" -delete
