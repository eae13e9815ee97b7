#!/bin/bash
# round 2, GPU call 47: plan construction through a pinned bounce buffer: first build of a process vs the next ones
OUT=gpurun_out/r02at; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "page_rank or pagerank or pb or hub or multi or smoke" > $OUT/pytest_pr.log 2>&1; grep -a "passed\|failed" $OUT/pytest_pr.log | tail -2
bash tools/runs/r02_call32.sh 2>&1 | grep -a "CALL\|hub flags\|hub groups" | head -12
GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 > $OUT/bench26.json 2> $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench26.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"
