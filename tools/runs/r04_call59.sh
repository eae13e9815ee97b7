#!/bin/bash
# round 4, GPU call 59: the accumulate kernel's phases (GM_PB_ABLATE 40 / 50 / 60 / 30: wrong results by design) at scale 26,
# under the tracer so that the kernel's own duration is read, then the final records on this library
OUT=gpurun_out/r04zzy; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python tools/ablate.py 26 40 50 60 30 > $OUT/ablate.log 2>&1; tail -12 $OUT/ablate.log
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/t/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = [(n, (e - s) / 1e3) for n, s, e in c.execute("select name, start, end from kernels order by start") if "pb_accum_kernel" in n]
by = {}
for n, d in rows:
    k = n.split("(")[0].split("pb_accum_kernel")[1]
    by.setdefault(k, []).append(d)
for k, v in by.items():
    v.sort(); print("pb_accum_kernel" + k, len(v), "dispatches, median", round(v[len(v) // 2], 1), "us")
PY
rm -rf $OUT/t
bash tools/runs/r04_final_c.sh
