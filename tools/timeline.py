#!/usr/bin/env python3
"""Start / end of every kernel dispatch of a few sweeps from the MIDDLE of a rocprofv3 --kernel-trace run (rocpd sqlite):
usage: timeline.py <dir-or-db> [sweeps=2]  ->  per dispatch: start relative to the sweep's first kernel, duration, kernel"""
import glob, os, sqlite3, sys

src, sweeps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
if "kernels" not in views:
    sys.exit(f"no 'kernels' view in {db}: {views}")
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
extra = [k for k in ("stream_id", "queue_id") if k in cols]
rows = list(c.execute(f"select name, start, end{''.join(', ' + k for k in extra)} from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "pb_bin_kernel" in r[0]]
if len(starts) < sweeps + 1:
    sys.exit("not enough sweeps in the trace")
mid = len(starts) // 2
lo, hi = starts[mid], starts[mid + sweeps] if mid + sweeps < len(starts) else len(rows)
t0 = rows[lo][1]
for r in rows[lo:hi]:
    if "pb_bin_kernel" in r[0]:
        print()
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{(r[1] - t0) / 1000.0:10.1f} us  +{(r[2] - r[1]) / 1000.0:8.1f} us  {' '.join(str(x) for x in r[3:])}  {name}")
