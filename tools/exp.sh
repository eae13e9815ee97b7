#!/bin/bash
# tools/exp.sh — experiment driver for a gpurun call: runs bench.py variants and prints a compact table.
# usage: tools/exp.sh "<scale> <relabel> <steps> <warmup> [extra args]" ...
for spec in "$@"; do
  set -- $spec
  scale=$1; relabel=$2; steps=$3; warm=$4; shift 4
  out=$(timeout 900 python bench.py --scale $scale --relabel $relabel --steps $steps --warmup $warm --cpu-sweeps 0 "$@" 2>&1 | tail -1)
  echo "$out" | python -c "
import sys, json
line = sys.stdin.read().strip()
try:
    d = json.loads(line)
    r = d['roofline']
    print('scale=$scale relabel=$relabel extra=[$*] engine=%s GTEPS=%.1f ms/step=%.4f tile_ms=%.4f frac=%.4f achieved=%.0fGB/s build=%.2fs' % (d['config']['engine'], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['achieved'], d['config']['csr_build_s']))
except Exception as e:
    print('scale=$scale relabel=$relabel FAILED:', line[-400:])
"
done
