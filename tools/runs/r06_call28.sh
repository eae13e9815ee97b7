#!/bin/bash
# round 6, GPU call 28: is the emulated rank's 40-50 us between a sweep's last kernel and the next sweep's first the host?
# (host time to enqueue a step beside the wall clock per step, with and without the event pairs around the pieces)
OUT=gpurun_out/r06aa; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], 'host enqueue', d['config'].get('host_enqueue_ms_per_step'), d['config']['value_stream_placement'].get('level'))"; }
for cfg in "8 0" "8 6" "4 0" "2 0"; do set -- $cfg
for ev in "" "--no-piece-events" "" "--no-piece-events"; do
timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts $1 --emulate-rank $2 $ev 2>> $OUT/bench.err | tail -1 | line "rank $2 of $1 [$ev]"
done; done
