#!/usr/bin/env python3
"""How many rows of an RMAT graph have many in-neighbours WITHOUT in-edges (sources whose out_score is the constant (1 - d) / n / out-degree):
per in-degree class, the largest number of such sources a row has — and, among them, the largest number that share one out-degree.
usage: leaf_sources_count.py <scale> [scale ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
for scale in [int(a) for a in sys.argv[1:]]:
    n = 1 << scale
    src, dst = synth.rmat_edges(scale, 42)
    indeg = torch.zeros(n, dtype=torch.int32, device=src.device); outdeg = torch.zeros_like(indeg)
    one = torch.ones(1 << 26, dtype=torch.int32, device=src.device)
    for lo in range(0, src.numel(), 1 << 26):
        hi = min(lo + (1 << 26), src.numel())
        indeg.index_add_(0, dst[lo:hi].long(), one[: hi - lo]); outdeg.index_add_(0, src[lo:hi].long(), one[: hi - lo])
    most = int(os.environ.get("LEAF_MAX_INDEG", "0"))  # sources with at most this many in-edges count (0: the plan's rule)
    leaf = indeg <= most
    cnt = torch.zeros(n, dtype=torch.int32, device=src.device)   # leaf sources per row
    cnt1 = torch.zeros(n, dtype=torch.int32, device=src.device)  # ... of out-degree 1 (the largest class of equal terms a row can have)
    for lo in range(0, src.numel(), 1 << 26):
        hi = min(lo + (1 << 26), src.numel())
        s = src[lo:hi].long(); d = dst[lo:hi].long()
        m = leaf[s]
        cnt.index_add_(0, d[m], one[: int(m.sum())])
        m1 = m & (outdeg[s] == 1)
        cnt1.index_add_(0, d[m1], one[: int(m1.sum())])
    print(f"scale {scale} (sources with <= {most} in-edges): {int(leaf.sum())} of {n} nodes; {int((leaf & (outdeg > 0)).sum())} of them have out-edges, "
          f"{int(outdeg[leaf].sum())} edges in all ({100.0 * int(outdeg[leaf].sum()) / src.numel():.1f} %)")
    for lo_d, hi_d in ((256, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 1 << 30)):
        sel = (indeg >= lo_d) & (indeg < hi_d)
        if int(sel.sum()) == 0:
            print(f"   in-degree [{lo_d}, {hi_d}): no rows"); continue
        c = cnt[sel]
        print(f"   in-degree [{lo_d}, {hi_d}): {int(sel.sum())} rows; sources without in-edges per row: max {int(c.max())}, mean {float(c.float().mean()):.1f}; rows with >= 512: {int((c >= 512).sum())}, "
              f">= 256: {int((c >= 256).sum())}; of out-degree 1: max {int(cnt1[sel].max())}")
