#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (rocpd sqlite):
usage: pmc_collect.py out.json <dir-or-db> [...]  -> {kernel: {counter: average value per dispatch, counter_total: sum over
the run's dispatches, counter_dispatches: how many}}"""
import glob, json, os, sqlite3, sys

out, srcs = sys.argv[1], sys.argv[2:]
res = {}
for src in srcs:
    dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    for db in dbs:
        c = sqlite3.connect(db)
        q = "select kernel_name, counter_name, avg(value), sum(value), count(*) from counters_collection group by kernel_name, counter_name"
        for kernel, counter, avg, total, cnt in c.execute(q):
            r = res.setdefault(kernel[:60], {})
            r[counter], r[counter + "_total"], r[counter + "_dispatches"] = avg, total, cnt
json.dump(res, open(out, "w"), indent=1)
print(f"{len(res)} kernels -> {out}")
