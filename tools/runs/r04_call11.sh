#!/bin/bash
# round 4, GPU call 11: one hub row of N terms alone on the GPU: duration of pb_hublong_kernel / pb_hubseq_kernel per dispatch
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
for kind in seq long; do
  timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/$kind -o t -- python tools/hub_probe.py $kind 9000 16384 131072 > $OUT/$kind.log 2>&1
  grep -a "N=" $OUT/$kind.log | head -3
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/$kind/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0] if any("kernel_dispatch" in t for t in tabs) else None
rows = list(c.execute("select name, start, end from kernels order by start")) if "kernels" in tabs else []
if not rows:
    print("tables:", tabs[:40])
cur = []
for name, s, e in rows:
    if "pb_hub" in name:
        cur.append((name.split("(")[0].split("::")[-1], (e - s) / 1000.0))
print("$kind:", " ".join(f"{n[:12]}={d:.1f}us" for n, d in cur))
PY
done
