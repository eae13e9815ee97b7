#!/bin/bash
# round 5, GPU call 38: after making the in-order schedule bench.py's default at every N: (1) bench.py --gpus 8 (gloo, one GPU, scale 26)
# with the defaults and with --piece-streams 1 (which now sets GPU_MAX_HW_QUEUES=16): final sweep error against one rank's
# 4.402272355163994e-05 after the same 25 sweeps; (2) emulated ranks, in order vs one stream per part, for DESIGN's table
OUT=gpurun_out/r05z; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
for ps in -1 1; do timeout 600 python bench.py --gpus 8 --backend gloo --single-device 1 --piece-streams $ps --cpu-sweeps 0 --algos 0 --prewarm-ms 0 --steps 20 --warmup 5 2> $OUT/gloo8_ps$ps.err | tail -1 > $OUT/gloo8_ps$ps.json
python -c "import json; d = json.loads(open('$OUT/gloo8_ps$ps.json').read()); c = d['config']; print('8 gloo ranks, scale 26, --piece-streams $ps:', c['final_sweep_error'], c['final_sweep_error'] == 4.402272355163994e-05, '|', c['partition'][-60:])" || tail -5 $OUT/gloo8_ps$ps.err; done
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'])"; }
for rep in 1 2; do for cfg in "8 0" "8 1" "4 0" "2 0"; do set -- $cfg; for ps in 0 1; do timeout 600 python bench.py --emulate-parts $1 --emulate-rank $2 --piece-streams $ps --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "emulated rank $2 of $1, piece-streams $ps"; done; done; done
