#!/usr/bin/env python3
"""How long do the hub kernels need for ONE row of N terms?  Star graphs (hubs rows x N sources, realistic term sizes), a few
sweeps each; run under `rocprofv3 --kernel-trace` and read the per-dispatch durations (tools/hub_probe.sh prints them).
usage: hub_probe.py <long|seq> N [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from graph_amd import prelude as P
from graph_amd.engine import PageRankEngine

kind = sys.argv[1]
os.environ["GM_PB_NOCACHE"] = "1"
os.environ["GM_PB_HUB_LONG"] = "4096" if kind == "long" else "1000000000"
hubs = int(os.environ.get("HUBS", "0")) or (1 if kind == "long" else 3)  # HUBS=<rows>: that many rows of N terms each
if "HUBS" in os.environ:
    os.environ.setdefault("GM_PB_HUB_GROUP", str(1 << 30))  # ... in one group (up to 64 rows)
for N in [int(a) for a in sys.argv[2:]]:
    n = hubs + N
    s = np.repeat(np.arange(hubs, n, dtype=np.uint32), hubs)
    d = np.tile(np.arange(hubs, dtype=np.uint32), N)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    od = np.bincount(s, minlength=n).astype(np.int32)
    eng = PageRankEngine(inc.handle, n, 0, torch.from_numpy(od).cuda(), 0.85, engine=PageRankEngine.PB)
    rng = np.random.default_rng(N)
    x0 = np.full(n, np.inf, np.float32)
    x0[hubs:] = (2.2e-9 / rng.integers(1, 40, N)).astype(np.float32)
    x_in = torch.from_numpy(x0).cuda()
    x_out = torch.empty_like(x_in)
    scores = torch.full((n,), 1.0 / n, device="cuda")
    err = torch.zeros(1, dtype=torch.float64, device="cuda")
    for _ in range(6):
        eng.sweep(x_in, x_out, scores, err)
    torch.cuda.synchronize()
    print(f"{kind} N={N}: done, info {eng.plan_info()['long_rows']} long rows", flush=True)
    del eng, inc
