#!/usr/bin/env python3
"""Per-DISPATCH counter values from rocprofv3 --pmc result databases (rocpd sqlite), in dispatch order:
usage: pmc_by_dispatch.py <dir-or-db> [kernel-substring ...]  ->  one line per dispatch: kernel, duration, counters"""
import glob, os, sqlite3, sys

src, pats = sys.argv[1], sys.argv[2:] or ["pb_bin_kernel", "pb_accum_kernel", "pb_hubseq_kernel", "pb_hublong_kernel"]
dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
for db in dbs:
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else "id"
    dur = "(end - start)" if "start" in cols and "end" in cols else "0"
    rows = {}
    for d, k, cn, v, t in c.execute(f"select {key}, kernel_name, counter_name, value, {dur} from counters_collection"):
        r = rows.setdefault(d, {"kernel": k, "dur_us": t / 1000.0})
        r.setdefault(cn, []).append(v)
    print(f"# {db}: columns {cols}")
    for d in sorted(rows):
        r = rows[d]
        if not any(p in r["kernel"] for p in pats):
            continue
        name = next(p for p in pats if p in r["kernel"])
        def show(v):  # one value, or the spread over the instances of a per-channel counter
            return f"{v[0]:.6g}" if len(v) == 1 else f"sum {sum(v):.6g} [{len(v)} inst: min {min(v):.4g} max {max(v):.4g} max/mean {max(v) * len(v) / max(sum(v), 1):.3f}]"
        extra = " ".join(f"{k}={show(r[k])}" for k in sorted(r) if k not in ("kernel", "dur_us"))
        print(f"{d:6d} {name:18s} {r['dur_us']:10.1f} us  {extra}")
