#!/bin/bash
# round 6, GPU call 6: WCC per-kernel times after the sample kernel's second change; SSSP per-round statistics; which of the
# transposed lists (in_edge: mask bit 4, in_off: bit 8) the arena fault follows now that in_off is filled by a scan
OUT=gpurun_out/r06f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/bench_algos.py --skip sssp,tc,prapi --oracle 2 --reps 5 > $OUT/wcc.json 2> $OUT/wcc.err; python -c "
import json; d=json.load(open('$OUT/wcc.json'))['wcc']; print('wcc', {k: d.get(k) for k in ('ms','best_ms','ms_result_left_on_device','best_ms_result_left_on_device','first_call_ms')}, d['parity'])"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/wcc_trace -o t -- python tools/bench_algos.py --profile 1 --skip sssp,tc,prapi > $OUT/wcc_record.json 2> $OUT/wcc_trace.err
DB=$(find $OUT/wcc_trace -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 12 | cut -c1-150
GM_SSSP_STATS=1 timeout 300 python - > $OUT/sssp_stats.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from graph_amd import synth, prelude as P
n = 1 << 24
src, dst = synth.rmat_edges(24, 42); w = synth.rmat_weights(int(src.numel()), 44)
go = synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted, w); g = P.DirectedCsrGraph(go, go, P.CsrLayout.Sorted)
import os
os.environ.pop('GM_SSSP_STATS', None)
for _ in range(2): P.delta_stepping(g, P.DeltaSteppingConfig(0, 0.1))
os.environ['GM_SSSP_STATS'] = '1'
P.delta_stepping(g, P.DeltaSteppingConfig(0, 0.1))
PY
grep -a "^sssp" $OUT/sssp_stats.txt | awk '{print}' | cut -c1-120 | head -80
for mask in 7 11; do
  GM_SSSP_ARENA=$mask GM_SSSP_TIMES=1 timeout 600 python bench.py --cpu-sweeps 2 --parity 0 --tc-oracle 0 > $OUT/bench_mask$mask.json 2> $OUT/bench_mask$mask.err
  echo "mask $mask rc=$? $(grep -ac 'Memory access fault' $OUT/bench_mask$mask.err) faults; $(grep -a '^sssp:' $OUT/bench_mask$mask.err | tr '\n' '|' | cut -c1-300)"
done
find $OUT -name "*.db" -delete
