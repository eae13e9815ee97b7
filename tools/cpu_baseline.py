#!/usr/bin/env python3
"""The CPU baseline of bench.py in isolation: the oracle's restatement of the reference's threaded
PageRank sweep on the host cores, with and without NUMA-spread inputs.  usage: cpu_baseline.py [scale] [sweeps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
from graph_amd import synth
from graph_amd._lib import check, lib, vp
from graph_amd.prelude import CsrLayout, Direction
from oracle import oracle as O
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
m = src.numel()
od = torch.bincount(src, minlength=n).to(torch.int32).cpu().numpy().astype(np.uint32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
off, tgt = np.empty(n + 1, np.uint32), np.empty(m, np.uint32)
check(lib().gm_csr_download(csr.handle, off.ctypes.data_as(vp), tgt.ctypes.data_as(vp), None))
del csr
threads = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [O.effective_cores()]
for cores in threads:
    for spread in (False, True):
        sec, err = O.page_rank_chunked_timed(off, tgt, od, sweeps, 0.85, cores, spread)
        print(f"scale {scale} threads {cores} spread={spread}: {sec / sweeps * 1e3:.1f} ms/sweep  {m * sweeps / sec / 1e9:.3f} GTEPS",
              flush=True)
