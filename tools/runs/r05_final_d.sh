#!/bin/bash
# round 5, final records (d), after the rank-major exchange layout of the one-process-per-GPU front: whole suite, the emulated
# partition table, an 8-way rank's timeline, the default line once more (both counter records in place)
OUT=gpurun_out/r05fd; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 2400 python tools/partition_emulated.py --scale 26 > $OUT/partition_emulated_scale26.json 2> $OUT/partition.err
python -c "
import json; d=json.load(open('$OUT/partition_emulated_scale26.json'))
for t in d['table']: print(t['gpus'], t['fastest_rank_ms'], t['slowest_rank_ms'], t['exchange_ms_model'], t['projected_sweep_ms'], t.get('projected_speedup'))"
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t8 -o t -- python bench.py --emulate-parts 8 --emulate-rank 0 --cpu-sweeps 0 --algos 0 > $OUT/t8.log 2>&1
python tools/timeline.py $OUT/t8 1 > $OUT/timeline_rank0_of_8.txt 2>&1; cut -c1-120 $OUT/timeline_rank0_of_8.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v['roofline'].get('frac'), v['roofline'].get('traffic')) for k, v in d['extra'].items() if isinstance(v, dict)})"
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --algos 0 2> $OUT/gloo2.err | tail -1 > $OUT/gloo2.json
find $OUT -name "*.db" -delete
