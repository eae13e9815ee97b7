#!/usr/bin/env python3
"""Per-CALL kernel sums and counters of `tools/bench_algos.py --profile 1` from rocprofv3 result databases (rocpd sqlite).

bench_algos in profile mode puts a marker launch (torch.cuda._sleep -> `spin_kernel`) in front of and behind every timed API
call and lists the segments' labels in its JSON record (`profile_segments`; "~" = between two calls), so the dispatches between
a call's two markers are that call:
WCC's single call, SSSP's first / second (builds the ordered lists) / third call on a handle, triangle count's first and
second call.

    algos_profile.py <record.json> <trace-dir-or-db> [<pmc-dir-or-db> ...]

Output (text): per segment the kernels of the library (name, dispatches, total us, average us), the span from the first
dispatch's start to the last one's end, and — from the --pmc databases, matched by the same markers — the counters summed
over the segment's dispatches per kernel.  FETCH_SIZE / WRITE_SIZE are printed as rocprofv3 reports them (KiB... see
tools/pmc_traffic.py for the gfx950 corrections: FETCH_SIZE of wide streaming reads is half the bytes)."""
import glob
import json
import os
import sqlite3
import sys

MARK = "spin_kernel"
OURS = ("wcc_", "sssp_", "tc_", "pb_", "pr_", "relabel_", "csr_", "gm_", "mg_")


def dbs_of(src):
    return [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def segments(rows, labels):
    """rows: [(name, ...)] in dispatch order -> [(label, [rows])] cut at the markers"""
    segs, cur, k = [], None, 0
    for r in rows:
        if MARK in r[0]:
            cur = []
            segs.append((labels[k] if k < len(labels) else f"segment {k}", cur))
            k += 1
        elif cur is not None:
            cur.append(r)
    return segs


def main():
    rec = json.load(open(sys.argv[1]))
    labels = rec.get("profile_segments", [])
    trace, pmcs = sys.argv[2], sys.argv[3:]
    out = {}
    for db in dbs_of(trace)[:1]:
        c = sqlite3.connect(db)
        rows = list(c.execute("select name, start, end from kernels order by start"))
        print(f"# kernel trace: {db}: {len(rows)} dispatches, {sum(1 for r in rows if MARK in r[0])} markers, labels {labels}")
        for label, seg in segments(rows, labels):
            mine = [r for r in seg if short(r[0]).startswith(OURS) or "rocprim" in r[0]]
            if not mine or label == "~":  # "~": between two calls (graph construction)
                continue
            span = (max(r[2] for r in mine) - min(r[1] for r in mine)) / 1e3
            busy = sum(r[2] - r[1] for r in mine) / 1e3
            print(f"\n## {label}: {len(mine)} dispatches, first start -> last end {span:.1f} us, sum of kernel durations {busy:.1f} us")
            agg = {}
            for name, s, e in mine:
                a = agg.setdefault(short(name), [0, 0.0])
                a[0] += 1
                a[1] += (e - s) / 1e3
            for name, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"   {name:70s} {cnt:6d} x {tot / cnt:10.2f} us = {tot:11.1f} us")
            out[label] = {"dispatches": len(mine), "span_us": span, "kernel_us": busy,
                          "kernels": {k: {"calls": v[0], "total_us": v[1]} for k, v in agg.items()}}
    for src in pmcs:
        for db in dbs_of(src):
            c = sqlite3.connect(db)
            rows = list(c.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection order by dispatch_id"))
            # one row per (dispatch, counter[, instance]): rebuild dispatch order, cut at the markers
            disp, order = {}, []
            for name, d, cn, v in rows:
                if d not in disp:
                    disp[d] = (name, {})
                    order.append(d)
                disp[d][1][cn] = disp[d][1].get(cn, 0.0) + v
            seq = [(disp[d][0], disp[d][1]) for d in order]
            counters = sorted({cn for _, cs in seq for cn in cs})
            print(f"\n# counters {counters}: {db}")
            for label, seg in segments(seq, labels):
                agg = {}
                for name, cs in seg:
                    if not (short(name).startswith(OURS)):
                        continue
                    a = agg.setdefault(short(name), {})
                    for cn, v in cs.items():
                        a[cn] = a.get(cn, 0.0) + v
                    a["dispatches"] = a.get("dispatches", 0) + 1
                if not agg or label == "~":
                    continue
                print(f"## {label}")
                for name, a in sorted(agg.items(), key=lambda kv: -max(v for k, v in kv[1].items() if k != "dispatches")):
                    print(f"   {name:60s} x{a['dispatches']:<5d} " + "  ".join(f"{cn}={a[cn]:.6g}" for cn in counters if cn in a))
                tot = {cn: sum(a.get(cn, 0.0) for a in agg.values()) for cn in counters}
                print("   " + " " * 60 + "  total  " + "  ".join(f"{cn}={tot[cn]:.6g}" for cn in counters))
                out.setdefault(label, {}).setdefault("counters", {}).update(tot)
    if os.environ.get("ALGOS_PROFILE_JSON"):
        json.dump(out, open(os.environ["ALGOS_PROFILE_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
