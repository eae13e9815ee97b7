#!/bin/bash
# round 3, GPU call 32: where the time of a sweep goes on the hub-owning slice of an 8-way partition: kernel timelines, PAR=0/1
OUT=gpurun_out/r03zb; mkdir -p $OUT; export TMPDIR=/tmp
for p in 0 1; do
cd /tmp && GM_PB_HUB_PAR=$p timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace$p -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 3 > $GRAFT_REPO_ROOT/$OUT/trace$p.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/trace$p 2 > $OUT/timeline_rank3_par$p.txt 2>&1; echo "== par $p"; cat $OUT/timeline_rank3_par$p.txt
find $OUT -name "*.db" -size +20M -delete
done
cd /tmp && GM_PB_HUB_PAR=1 timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/tracef -o trace -- python $GRAFT_REPO_ROOT/bench.py --cpu-sweeps 0 > $GRAFT_REPO_ROOT/$OUT/tracef.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/timeline.py $OUT/tracef 1 > $OUT/timeline_full_par1.txt 2>&1; echo "== full par 1"; cat $OUT/timeline_full_par1.txt
find $OUT -name "*.db" -size +20M -delete
