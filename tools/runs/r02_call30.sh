#!/bin/bash
# round 2, GPU call 30: allocation costs, plan phases, the whole GPU suite after the gm_csr / SSSP changes
OUT=gpurun_out/r02ae; mkdir -p $OUT; export TMPDIR=/tmp
./tools/allocbench > $OUT/allocbench.txt 2>&1; cat $OUT/allocbench.txt
GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 > $OUT/bench26.json 2> $OUT/bench26.err; grep "pb plan" $OUT/bench26.err; python -c "
import json; d=json.loads(open('$OUT/bench26.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['plan_build_ms'])"
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
