#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE / L2 misses of tools/fetchcal.hip's kernels against the bytes each of them requests.
usage: fetchcal_report.py <dir with rocprofv3 --pmc result databases> -> one line per kernel and counter (last dispatch of each kernel)"""
import glob, json, os, sqlite3, sys

KNOWN = {  # kernel -> (accesses, bytes requested per access, what a whole 128-byte line per access would be)
    "cal_stream16": (1 << 30, 1, None), "cal_stream8": (1 << 30, 1, None), "cal_rec128": (64 << 20, 128, 128),
    "cal_gather4<false>": (256 << 20, 4, 128), "cal_gather4<true>": (256 << 20, 4, 128), "cal_write4": (256 << 20, 4, 128),
}
rows = {}
for db in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else "id"
    for d, k, cn, v in c.execute(f"select {key}, kernel_name, counter_name, value from counters_collection order by {key}"):
        name = next((n for n in KNOWN if n.split("<")[0] + ("<" if "<" in n else "(") in k.replace(" ", "") and (("<" not in n) or ("<" + n.split("<")[1] in k.replace(" ", "")))), None)
        if name:
            rows.setdefault(name, {}).setdefault(cn, {}).setdefault(d, 0.0)
            rows[name][cn][d] += v
out = {}
for name, (count, per, line) in KNOWN.items():
    for cn, by_dispatch in sorted(rows.get(name, {}).items()):
        v = by_dispatch[max(by_dispatch)]  # the second (warm) launch
        req = count * per
        rec = {"value": v, "requested_bytes": req}
        if cn in ("FETCH_SIZE", "WRITE_SIZE"):  # reported in KiB
            b = v * 1024.0
            rec.update({"bytes_reported": b, "reported_over_requested": b / req, "reported_per_access": b / count})
            if line:
                rec["reported_over_whole_lines"] = b / (count * line)
            print(f"{name:20s} {cn:12s} {b / 1e9:9.3f} GB reported for {req / 1e9:8.3f} GB requested: x{b / req:.3f}" +
                  (f", {b / count:.1f} B per access ({b / (count * line):.3f} of a 128-byte line each)" if line else ""))
        else:
            rec["per_access"] = v / count
            print(f"{name:20s} {cn:12s} {v:14.0f} = {v / count:.3f} per access" if line else f"{name:20s} {cn:12s} {v:14.0f} = {v * 128 / req:.3f} x (bytes / 128)")
        out.setdefault(name, {})[cn] = rec
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
