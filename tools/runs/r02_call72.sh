#!/bin/bash
# round 2, GPU call 72: the self-launching multi-rank path of bench.py (two gloo ranks on one GPU) after this round's edits
OUT=gpurun_out/r02bs; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 > $OUT/two_ranks.json 2> $OUT/two_ranks.err; tail -c 700 $OUT/two_ranks.json; tail -3 $OUT/two_ranks.err
timeout 600 python bench.py --emulate-parts 8 --emulate-rank 0 --cpu-sweeps 0 > $OUT/emu8.json 2> $OUT/emu8.err; python -c "
import json; d=json.loads(open('$OUT/emu8.json').read().strip().splitlines()[-1]); print('emulated rank 0 of 8:', d['ms_per_step'], d['config'].get('emulated'))"
