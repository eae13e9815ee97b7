#!/bin/bash
# round 5, final records (c): every rank's slice of a 1 / 2 / 4 / 8-way partition of scale 26 timed alone (EMULATION), the hub
# threshold A/B in alternating processes, the accumulate phases' ablation on the measurement library
OUT=gpurun_out/r05fc; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python tools/partition_emulated.py --scale 26 > $OUT/partition_emulated_scale26.json 2> $OUT/partition.err; grep -a "parts" $OUT/partition.err | tr '\n' ' '; echo
python -c "
import json; d=json.load(open('$OUT/partition_emulated_scale26.json'))
for t in d['table']: print(t['gpus'], t['fastest_rank_ms'], t['slowest_rank_ms'], t['exchange_ms_model'], t['projected_sweep_ms'], t.get('projected_speedup'))"
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('$1:', d['ms_per_step'], d['roofline']['frac'], c['value_stream_placement'].get('level'), c['hub_rows_in_reference_order']['hub_rows'])"; }
for rep in 1 2 3; do for deg in 4096 1024; do
  GM_PB_HUB_DEG=$deg timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 26 hub_deg $deg"
done; done > $OUT/hub_deg_ab.txt; cat $OUT/hub_deg_ab.txt
for rep in 1 2 3; do for deg in 4096 1024; do
  GM_PB_HUB_DEG=$deg timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | line "scale 22 hub_deg $deg"
done; done >> $OUT/hub_deg_ab.txt; tail -6 $OUT/hub_deg_ab.txt
timeout 900 python tools/ablate.py 26 50 60 > $OUT/ablate.txt 2>&1; tail -2 $OUT/ablate.txt
rm -f graph_amd/libgraph_mi355x_measure.so
