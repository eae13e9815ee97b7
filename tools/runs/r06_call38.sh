#!/bin/bash
# round 6, GPU call 38: the hub threshold on the final kernels, parity margin against speed at the three BASELINE sizes:
# GM_PB_HUB_DEG = 4096 (default) / 8192 / 16384 / 32768 with the in-run parity leg (every row against the oracle's threaded path)
OUT=gpurun_out/r06ak; mkdir -p $OUT; export TMPDIR=/tmp
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; p = d['config'].get('parity') or {}; print('$1', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'hub rows', h.get('hub_rows'), 'hub edges', h.get('hub_edges'), 'parity', p.get('max_rel_vs_reference'), p.get('rows_over_1e-5'))"; }
for s in 22 24; do for hd in 4096 8192 16384 32768; do
GM_PB_HUB_DEG=$hd timeout 600 python bench.py --scale $s --algos 0 --cpu-sweeps 2 2>> $OUT/bench.err | tail -1 | line "scale $s hub_deg=$hd"
done; done
GM_PB_HUB_DEG=32768 timeout 600 python bench.py --algos 0 --cpu-sweeps 2 2>> $OUT/bench.err | tail -1 | line "scale 26 hub_deg=32768"
for hd in 16384 4096 16384 4096; do
GM_PB_HUB_DEG=$hd timeout 600 python bench.py --algos 0 --cpu-sweeps 0 2>> $OUT/bench.err | tail -1 | line "scale 26 hub_deg=$hd"
done
