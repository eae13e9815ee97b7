#!/bin/bash
# round 4, GPU call 39: SSSP scale 24 with weight-ordered lists: every dispatch of one call
OUT=gpurun_out/r04zh; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,tc > $OUT/kt.log 2>&1
python - <<PY
import sqlite3, glob, re
db = glob.glob("$OUT/kt/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels order by start") if "sssp" in n]
# the last call only: from the last sssp_init_kernel on
last = max(i for i, r in enumerate(rows) if "sssp_init" in r[0])
rows = rows[last:]
t0 = rows[0][1]
tot = {}
out = []
for n, s, e in rows:
    k = re.search(r"sssp_\w+", n).group(0)
    d = (e - s) / 1e3
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += d
    out.append(f"{k:24s} {(s - t0) / 1e3:9.1f} +{d:8.1f}")
open("$OUT/sssp_dispatches.txt", "w").write("\n".join(out))
print({k: (c, round(t)) for k, (c, t) in tot.items()}, "span us", round((rows[-1][2] - t0) / 1e3))
big = sorted(((e - s) / 1e3, re.search(r"sssp_\w+", n).group(0), (s - t0) / 1e3) for n, s, e in rows)[-14:]
print([(round(d), k[5:10], round(at)) for d, k, at in big])
PY
rm -rf $OUT/kt
