#!/bin/bash
# round 6, GPU call 1: the torch-free reproducer of the stream-per-part schedule (tools/streams_repro.hip), 8 processes on one GPU,
# every knob of the event graph one at a time; the round-5 tool once more to see the failure on today's box; a baseline bench line
OUT=gpurun_out/r06a; mkdir -p $OUT; export TMPDIR=/tmp; export OMP_NUM_THREADS=1
R=tools/streams_repro
rep() { timeout 120 $R "$@" 2>&1 | tail -9 >> $OUT/repro.txt; echo "   rc=$? :: $*" >> $OUT/repro.txt; }
for i in 1 2 3 4; do rep --procs 8 --sweeps 300; done
for i in 1 2; do rep --procs 8 --sweeps 300 --streams 0; done
for v in "--null-main 0" "--fresh-events 0" "--side-prio 0" "--fork 0" "--thread 0" "--lockstep 0" "--mb 256" "--chain 512" "--parts 4"; do
  for i in 1 2; do rep --procs 8 --sweeps 300 $v; done
done
for i in 1 2; do GPU_MAX_HW_QUEUES=16 rep --procs 8 --sweeps 300; done
for i in 1 2; do rep --procs 16 --sweeps 200; done
for i in 1 2; do rep --procs 1 --sweeps 300; done
grep "streams_repro procs" $OUT/repro.txt
run() { local w=$1 s=$2; shift 2; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) tools/debug_multi_gloo.py --scale $s "$@" 2>> $OUT/debug.err | grep "^{" | tee -a $OUT/debug_multi_gloo.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('world', 'streams', 'gather', 'rows_that_differ', 'first_sweep_whose_error_differs')}, d['first_sweep_whose_scores_differ'])"; }
for i in 1 2 3; do run 8 22 --streams 1 --snap 1 --sweeps 20; done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
