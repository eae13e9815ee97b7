#!/usr/bin/env python3
"""Dump the per-kernel summary of a rocprofv3 rocpd (.db) result as text (the same numbers
`rocprofv3 --kernel-trace --stats` prints): name, calls, total / average duration in us, share."""
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) < 110 else name[:107] + "..."


def main(path, top=25):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: {path}")
    print(f"{'kernel':110s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, total, avg, pct in rows[:top]:
        print(f"{short(name):110s} {calls:7d} {total:14.3f} {avg:12.3f} {pct:7.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
