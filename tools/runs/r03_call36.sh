#!/bin/bash
# round 3, GPU call 36: what a sweep at scale 22 waits for: kernel timelines, long chains block-parallel or not
OUT=gpurun_out/r03zd; mkdir -p $OUT; export TMPDIR=/tmp
for p in 1 0; do
cd /tmp && GM_PB_HUB_PAR=$p timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/tl$p -o trace -- python $GRAFT_REPO_ROOT/bench.py --scale 22 --cpu-sweeps 0 --steps 6 --warmup 2 --prewarm-ms 0 > $GRAFT_REPO_ROOT/$OUT/tl$p.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/tl$p 2 > $OUT/sweep_timeline_scale22_par$p.txt 2>&1; echo "== par $p"; cat $OUT/sweep_timeline_scale22_par$p.txt
done
find $OUT -name "*.db" -delete
