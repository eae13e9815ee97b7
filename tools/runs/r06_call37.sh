#!/bin/bash
# round 6, GPU call 37: fuzz + stress on the library in the tree (c03c6d43), then the hub threshold once more on the final kernels:
# GM_PB_HUB_DEG = 4096 (default) / 8192 / 16384 at scale 26 with the in-run parity leg (margin against 1e-5) and at scale 22
OUT=gpurun_out/r06aj; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/runs/r06_call16.sh > $OUT/fuzz.log 2>&1; grep -a "cases\|IDENTICAL" gpurun_out/r06o/*.log | cut -c1-160
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; p = d['config'].get('parity') or {}; print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'hub rows', h.get('hub_rows'), 'hub edges', h.get('hub_edges'), 'parity', p.get('max_rel_vs_reference'), p.get('rows_over_1e-5'))"; }
for hd in 4096 8192 16384 4096 8192; do
GM_PB_HUB_DEG=$hd timeout 600 python bench.py --algos 0 --cpu-sweeps 2 2>> $OUT/bench.err | tail -1 | line "scale 26 hub_deg=$hd"
done
for hd in 4096 8192 4096 8192; do
GM_PB_HUB_DEG=$hd timeout 300 python bench.py --scale 22 --algos 0 --cpu-sweeps 0 2>> $OUT/bench.err | tail -1 | line "scale 22 hub_deg=$hd"
done
