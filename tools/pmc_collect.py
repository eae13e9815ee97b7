#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc result databases (rocpd sqlite):
usage: pmc_collect.py out.json <dir-or-db> [...]  -> {kernel: {counter: average value per dispatch}}"""
import glob, json, os, sqlite3, sys

out, srcs = sys.argv[1], sys.argv[2:]
res = {}
for src in srcs:
    dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    for db in dbs:
        c = sqlite3.connect(db)
        q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
        for kernel, counter, avg, cnt in c.execute(q):
            res.setdefault(kernel[:60], {})[counter] = avg
json.dump(res, open(out, "w"), indent=1)
print(f"{len(res)} kernels -> {out}")
