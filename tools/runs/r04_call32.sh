#!/bin/bash
# round 4, GPU call 32: pb_hubseq_kernel, one group alone on the GPU: time per block of 2040 terms against the rows of the group
OUT=gpurun_out/r04za; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in "64 4096" "32 8192" "16 16384" "8 32768" "4 65536" "2 131072"; do
  set -- $cfg
  HUBS=$1 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python tools/hub_probe.py seq $2 > $OUT/p.log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/p/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = [(n, (e - s) / 1000.0) for n, s, e in c.execute("select name, start, end from kernels order by start") if "pb_hubseq_kernel" in n]
d = sorted(x[1] for x in rows)[len(rows) // 2]
print(f"rows $1 x $2 terms: {d:.1f} us per dispatch, {d / ($1 * $2 / 2040.0):.2f} us per block, {d * 1e3 / ($1 * $2):.2f} ns per term of the group, {d * 1e3 / $2:.2f} ns per term of a row")
PY
  rm -rf $OUT/p
done
