#!/usr/bin/env python3
"""PageRankConfig::default() (20 iterations, tolerance 1e-4; crates/algos/src/page_rank.rs:14-56) on the device against the
reference's threaded path: the device runs synchronous (Jacobi) sweeps, the reference updates out_scores in place
(page_rank.rs:155-159: Gauss-Seidel-like inside a chunk), so the reference reaches the tolerance in fewer iterations.
Records iterations, error and the row-wise difference of the two (unconverged) results (INTEGRATION.md "what differs").

    python tools/default_config_gap.py --scale 22"""
import argparse, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
args = ap.parse_args()
import numpy as np

from graph_amd import prelude as P
from graph_amd import synth
from oracle import oracle as O  # the checker

sc, n = args.scale, 1 << args.scale
src, dst = synth.rmat_edges(sc, 42)
g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                       synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src, dst
got, it_g, err_g = P.page_rank(g, P.PageRankConfig())
ioff, itgt, _ = g.csr_inc.host()
od = g.csr_out.degrees().astype(np.uint32)
ref, it_r, err_r = O.page_rank_chunked(ioff, itgt, od, 20, 1e-4, 0.85, O.effective_cores())
rel = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)
# the same tolerance reached by the device when it is given the iterations: how far apart are the two STOPPED results
got2, it2, err2 = P.page_rank(g, P.PageRankConfig(200, 1e-4, 0.85))
rel2 = np.abs(got2.astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)
print(json.dumps({"tool": "default_config_gap", "scale": sc,
                  "device_default": {"iterations": int(it_g), "error": float(err_g)},
                  "reference_default": {"iterations": int(it_r), "error": float(err_r)},
                  "max_rel_device_vs_reference": float(rel.max()), "median_rel": float(np.median(rel)),
                  "l1_difference": float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).sum()),
                  "device_with_200_iterations_allowed": {"iterations": int(it2), "error": float(err2),
                                                         "max_rel_vs_reference_default": float(rel2.max())}}))
