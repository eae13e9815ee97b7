#!/bin/bash
# round 2, GPU call 79: per-sweep time over 12 s inside one process, three processes (does the level move in time?)
OUT=gpurun_out/r02bz; mkdir -p $OUT; export TMPDIR=/tmp
for k in 1 2 3; do timeout 300 python tools/timeseries.py 26 4000 > $OUT/ts$k.txt 2>&1; python - <<PY
import re
v=[float(m.group(1)) for m in re.finditer(r'ms/sweep=([\d.]+)', open('$OUT/ts$k.txt').read())]
print('process $k: blocks', len(v), 'min', min(v), 'max', max(v), 'first 5', v[:5], 'last 5', v[-5:])
PY
done
