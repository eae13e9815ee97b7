#!/bin/bash
# round 2, GPU call 73: every rank's slice of a 2- / 4- / 8-way split of the scale-26 graph, one at a time on one GPU
# (kernel time of the partitioned sweep per rank, the exchange replaced by a local copy): the balance of the partition
OUT=gpurun_out/r02bt; mkdir -p $OUT; export TMPDIR=/tmp
for P in 2 4 8; do
  for ((r=0; r<P; r++)); do
    timeout 600 python bench.py --emulate-parts $P --emulate-rank $r --cpu-sweeps 0 > $OUT/emu_${P}_$r.json 2>/dev/null
    python -c "
import json; d=json.loads(open('$OUT/emu_${P}_$r.json').read().strip().splitlines()[-1]); print('parts $P rank $r ms', d['ms_per_step'], 'kernels', d['roofline']['avg_launch_ms'], 'edges', d['roofline']['edges_per_launch'], 'rows', d['roofline']['rows_per_launch'])"
  done
done
