#!/usr/bin/env python3
"""Differential fuzz of the HIP path against the oracle on random small graphs of awkward shapes (duplicates,
self-loops, isolated nodes, hubs, empty graphs): CSR build in every direction and layout, PageRank (sequential
mode bit-exact; PB and pull engines against exact row sums), WCC ids, SSSP distances (zero weights included),
triangle counts with and without relabelling.  usage: fuzz_parity.py [cases] [seed] [max nodes] [max edges];
exit status 1 on a mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
# the PageRank check below compares against EXACT row sums: rows with >= 4096 in-edges would otherwise follow the
# reference's left-to-right f32 order (a ~1e-4 difference on hub rows that is the reference's, not a bug)
os.environ.setdefault("GM_PB_HUB_DEG", "0")
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
max_m = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
rng = np.random.default_rng(seed)
bad = 0
quirks = 0


def fail(case, what):
    global bad
    bad += 1
    print(f"MISMATCH case {case}: {what}", flush=True)


def exact_sweeps(ioff, itgt, od, sweeps, damping=0.85):
    n = ioff.size - 1
    init = np.float32(1.0) / np.float32(n)
    base = (np.float32(1.0) - np.float32(damping)) / np.float32(n)
    scores = np.full(n, init, np.float32)
    odf = od.astype(np.float32)
    with np.errstate(divide="ignore"):
        outs = (init / odf).astype(np.float32)
    row = np.repeat(np.arange(n), np.diff(ioff).astype(np.int64))
    for _ in range(sweeps):
        inc = np.bincount(row, weights=outs[itgt].astype(np.float64), minlength=n).astype(np.float32)
        scores = (base + (np.float32(damping) * inc).astype(np.float32)).astype(np.float32)
        with np.errstate(divide="ignore"):
            outs = (scores / odf).astype(np.float32)
    return scores


for case in range(cases):
    shape = rng.integers(0, 5)
    n = int(rng.integers(1, max_n))
    m = int(rng.integers(0, max_m)) if shape else 0
    s = rng.integers(0, n, m).astype(np.uint32)
    d = rng.integers(0, n, m).astype(np.uint32)
    if shape == 2 and m:  # a hub on each side
        s[: m // 3] = rng.integers(0, n)
        d[m // 3: 2 * m // 3] = rng.integers(0, n)
    if shape == 3 and m:  # few distinct endpoints: many duplicates and self-loops
        k = int(rng.integers(1, 12))
        s, d = (s % k).astype(np.uint32), (d % k).astype(np.uint32)
    if shape == 4 and m:  # only the lower half of the ids is used
        s, d = (s // 2).astype(np.uint32), (d // 2).astype(np.uint32)
    w = rng.choice(np.array([0.0, 0.125, 0.5, 1.0, 2.75], np.float32), m) if m else np.zeros(0, np.float32)
    layout = int(rng.integers(0, 3))
    tag = f"n={n} m={m} shape={shape} layout={layout}"
    # CSR build
    for direction in (O.OUTGOING, O.INCOMING, O.UNDIRECTED):
        ref = O.csr_build(n, s, d, direction, layout, w if layout != O.DEDUPLICATED else None)
        got = P.DeviceCsr.from_edges(n, s, d, w if layout != O.DEDUPLICATED else None, direction, layout).host()
        if not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
            fail(case, f"csr_build dir {direction} {tag}")
    slayout = O.SORTED if layout == O.UNSORTED else layout  # algorithms below want a defined neighbour order
    out = P.DeviceCsr.from_edges(n, s, d, w if slayout == O.SORTED else None, P.Direction.Outgoing, slayout)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, slayout)
    g = P.DirectedCsrGraph(out, inc, P.CsrLayout(slayout))
    ooff, otgt = O.csr_build(n, s, d, O.OUTGOING, slayout)[:2]
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, slayout)[:2]
    od = np.diff(ooff).astype(np.uint32)
    # PageRank: Auto = the reference's sequential order for n <= 16384, bit for bit
    ref = O.page_rank_seq(ioff, itgt, od, 20, 1e-4, 0.85)
    got = P.page_rank(g, P.PageRankConfig(20, 1e-4, 0.85), P.PageRankMode.Sequential)
    if not (np.array_equal(got[0], ref[0]) and got[1] == ref[1] and got[2] == ref[2]):
        fail(case, f"page_rank sequential {tag}")
    ex = exact_sweeps(ioff, itgt, od, 3)
    for mode in (P.PageRankMode.JacobiPB, P.PageRankMode.JacobiPull):
        got = P.page_rank(g, P.PageRankConfig(3, 0.0, 0.85), mode)
        tol = 1.5e-7 if mode == P.PageRankMode.JacobiPB else 5e-6
        if got[1] != 3 or not np.allclose(got[0], ex, rtol=tol, atol=0):
            fail(case, f"page_rank {mode.name} {tag}: max rel {np.max(np.abs(got[0] - ex) / ex):.3g}")
    # WCC
    if not np.array_equal(P.wcc_afforest(g).to_vec(), O.wcc(ooff, otgt, ioff, itgt)):
        fail(case, f"wcc {tag}")
    if not np.array_equal(P.wcc_baseline(g).to_vec(), O.wcc(ooff, otgt, ioff, itgt)):
        fail(case, f"wcc_baseline {tag}")
    # SSSP on the Sorted weighted out-CSR
    if slayout == O.SORTED:
        off_w, tgt_w, w_w = O.csr_build(n, s, d, O.OUTGOING, O.SORTED, w)
        start = int(rng.integers(0, n))
        delta = float(rng.choice([0.05, 0.3, 3.0]))
        got = P.delta_stepping(g, P.DeltaSteppingConfig(start, delta))
        fp = O.sssp_fixed_point(off_w, tgt_w, w_w, start)
        if not np.array_equal(got, fp):
            fail(case, f"sssp (least fixed point) start {start} delta {delta} {tag}")
        if O.stale_check_misfires(fp, delta).any():
            quirks += 1  # the reference's stale check drops an update on this input: its result is not the fixed point
        elif not np.array_equal(got, O.delta_stepping(off_w, tgt_w, w_w, start, delta)):
            fail(case, f"sssp (reference order) start {start} delta {delta} {tag}")
    # triangle count (put-back semantics on Sorted lists with duplicates and self-loops, or Deduplicated)
    ulayout = slayout
    ug = P.UndirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, 2, ulayout), P.CsrLayout(ulayout))
    uoff, utgt = O.csr_build(n, s, d, O.UNDIRECTED, ulayout)[:2]
    if P.global_triangle_count(ug) != O.triangle_count(uoff, utgt):
        fail(case, f"triangle_count {tag}")
    roff, rtgt, new_id = O.relabel_by_degree(uoff, utgt)
    got_id = ug.make_degree_ordered()
    if not np.array_equal(got_id, new_id) or P.global_triangle_count(ug) != O.triangle_count(roff, rtgt):
        fail(case, f"relabel + triangle_count {tag}")
print(f"{cases} cases, {bad} mismatches ({quirks} SSSP inputs on which the reference's stale check misfires)")
sys.exit(1 if bad else 0)
