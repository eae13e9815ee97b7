"""Pins the CPU oracle (oracle/graph_oracle.c) to the reference's own golden vectors.

Every expected value below is copied from a reference test / doc-test (file:line cited);
values marked "derived" are not reference-published and only guard against regressions.
"""
import heapq
import os

import numpy as np
import pytest

# README / doc-test graph: crates/algos/src/lib.rs:96-117
README_EDGES = [(1, 2), (2, 1), (4, 0), (4, 1), (5, 4), (5, 1), (5, 6), (6, 1), (6, 5), (7, 1), (7, 5),
                (8, 1), (8, 5), (9, 1), (9, 5), (10, 1), (10, 5), (11, 5), (12, 5)]
# crates/algos/src/lib.rs:121-139 — assert_eq! on f32
README_SCORES = np.array([0.024064068, 0.3145448, 0.27890152, 0.01153846, 0.029471997, 0.06329483,
                          0.029471997] + [0.01153846] * 6, np.float32)


def _edges(lst):
    return (np.array([e[0] for e in lst], np.uint32), np.array([e[1] for e in lst], np.uint32))


def test_page_rank_readme_vector_bit_exact(oracle):
    s, d = _edges(README_EDGES)
    n = 13
    for layout in (oracle.UNSORTED, oracle.SORTED):
        ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, layout)
        scores, iters, _ = oracle.page_rank_seq(ioff, itgt, oracle.out_degrees_from(n, s), 10, 1e-4, 0.85)
        assert iters == 10
        assert np.array_equal(scores, README_SCORES)


def test_page_rank_two_components_bit_exact(oracle):
    # crates/algos/src/page_rank.rs:175-197: GDL (a)-->()-->()<--(a),(b)-->()-->()<--(b), Sorted, defaults
    s, d = _edges([(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)])
    ioff, itgt = oracle.csr_build(6, s, d, oracle.INCOMING, oracle.SORTED)
    scores, _, _ = oracle.page_rank_seq(ioff, itgt, oracle.out_degrees_from(6, s))
    expected = np.array([0.024999997, 0.035624996, 0.06590624] * 2, np.float32)
    assert np.array_equal(scores, expected)


def test_page_rank_behaviours_scale8(oracle, scale8):
    # crates/mate/tests/page_rank_test.py:19-33
    s, d, n = scale8
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    assert oracle.page_rank_seq(ioff, itgt, od, max_iterations=1)[1] == 1
    assert oracle.page_rank_seq(ioff, itgt, od, tolerance=1.0)[1] == 1
    scores, iters, _ = oracle.page_rank_seq(ioff, itgt, od, damping=0.0)
    assert iters == 1
    assert np.all(scores == np.float32(1.0) / np.float32(256))
    # derived: default config
    scores, iters, err = oracle.page_rank_seq(ioff, itgt, od)
    assert iters == 7 and abs(err - 5.0109e-05) < 1e-8
    # the threaded path with one chunk (n <= 16384) is the sequential path
    sc2, it2, err2 = oracle.page_rank_chunked(ioff, itgt, od, threads=4)
    assert np.array_equal(sc2, scores) and it2 == iters and err2 == err


def test_page_rank_example_el(oracle, golden_dir):
    # BASELINE config 0: resources/example.el, page_rank(10, 1e-4, 0.85); values derived
    s, d = oracle.read_edge_list(os.path.join(golden_dir, "example.el"))
    n = int(max(s.max(), d.max())) + 1
    assert (n, s.size) == (4, 5)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    ooff, otgt = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED)
    assert list(otgt[ooff[1]:ooff[2]]) == [2, 3] and list(itgt[ioff[1]:ioff[2]]) == [0]  # builder/src/lib.rs:104-130
    scores, iters, err = oracle.page_rank_seq(ioff, itgt, oracle.out_degrees_from(n, s), 10, 1e-4, 0.85)
    assert iters == 2 and err == 0.0
    assert np.array_equal(scores, np.array([0.037499994, 0.053437494, 0.07614843, 0.124937095], np.float32))


def test_csr_build_scale8(oracle, scale8):
    # crates/builder/tests/builder.rs:448-491
    s, d, n = scale8
    assert n == 256 and s.size == 4096
    ooff, otgt = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    assert list(otgt[ooff[0]:ooff[1]]) == [37, 157]
    assert list(itgt[ioff[0]:ioff[1]]) == [12, 26, 50, 50, 52, 82, 82, 82, 106, 109, 172, 186, 250, 250]
    uoff, utgt = oracle.csr_build(n, s, d, oracle.UNDIRECTED, oracle.SORTED)
    assert uoff[1] - uoff[0] == 16 and utgt.size == 8192


def test_csr_layouts_small(oracle):
    # crates/builder/src/graph/csr.rs:998-1043 style: sort / dedup / self-loop removal
    s, d = _edges([(0, 3), (0, 1), (0, 1), (0, 0), (1, 2), (2, 2), (2, 0), (2, 0)])
    off, tgt = oracle.csr_build(4, s, d, oracle.OUTGOING, oracle.UNSORTED)
    assert list(off) == [0, 4, 5, 8] + [8] and list(tgt) == [3, 1, 1, 0, 2, 2, 0, 0]
    off, tgt = oracle.csr_build(4, s, d, oracle.OUTGOING, oracle.SORTED)
    assert list(tgt) == [0, 1, 1, 3, 2, 0, 0, 2]
    off, tgt = oracle.csr_build(4, s, d, oracle.OUTGOING, oracle.DEDUPLICATED)
    assert list(off) == [0, 2, 3, 4, 4] and list(tgt) == [1, 3, 2, 0]
    # undirected: out-direction entries first, then in-direction (csr.rs:154-172)
    s, d = _edges([(0, 1), (2, 0)])
    off, tgt = oracle.csr_build(3, s, d, oracle.UNDIRECTED, oracle.UNSORTED)
    assert list(off) == [0, 2, 3, 4] and list(tgt) == [1, 2, 0, 0]


def test_relabel_mapping(oracle):
    # crates/builder/src/graph_ops.rs:709-774: degrees (3,4,5,4) -> pairs [(5,2),(4,3),(4,1),(3,0)] -> [3,2,0,1]
    off = np.array([0, 3, 7, 12, 16], np.uint32)
    tgt = np.zeros(16, np.uint32)
    _, _, new_id = oracle.relabel_by_degree(off, tgt)
    assert list(new_id) == [3, 2, 0, 1]


def test_triangle_count_goldens(oracle, scale8):
    s, d, n = scale8
    uoff, utgt = oracle.csr_build(n, s, d, oracle.UNDIRECTED, oracle.SORTED)
    # crates/mate/tests/triangle_count_test.py:5-9 (after test_reorder mutated the shared fixture)
    roff, rtgt, _ = oracle.relabel_by_degree(uoff, utgt)
    assert oracle.triangle_count(roff, rtgt) == 227874
    assert oracle.triangle_count(roff, rtgt, threads=4) == 227874
    # derived: un-relabelled Sorted, and Deduplicated (true triangle count, relabel-invariant)
    assert oracle.triangle_count(uoff, utgt) == 256533
    doff, dtgt = oracle.csr_build(n, s, d, oracle.UNDIRECTED, oracle.DEDUPLICATED)
    assert oracle.triangle_count(doff, dtgt) == 10508
    r2 = oracle.relabel_by_degree(doff, dtgt)
    assert oracle.triangle_count(r2[0], r2[1]) == 10508


@pytest.mark.parametrize("edges", [
    # crates/algos/src/triangle_count.rs:93-130, crates/mate/tests/triangle_count_test.py:12-77
    [(0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 3)],
    [(0, 1), (1, 2), (2, 0), (0, 3), (3, 4), (4, 0)],
    [(0, 1), (1, 2), (2, 0), (1, 3), (3, 2)],
])
def test_triangle_count_shapes(oracle, edges):
    s, d = _edges(edges)
    n = int(max(s.max(), d.max())) + 1
    off, tgt = oracle.csr_build(n, s, d, oracle.UNDIRECTED, oracle.DEDUPLICATED)
    assert oracle.triangle_count(off, tgt) == 2


def _dijkstra_f32(off, tgt, w, start):
    n = off.size - 1
    dist = np.full(n, np.finfo(np.float32).max, np.float32)
    dist[start] = 0
    pq = [(np.float32(0), start)]
    while pq:
        du, u = heapq.heappop(pq)
        if du > dist[u]:
            continue
        for i in range(off[u], off[u + 1]):
            nd = np.float32(du + w[i])
            if nd < dist[tgt[i]]:
                dist[tgt[i]] = nd
                heapq.heappush(pq, (nd, int(tgt[i])))
    return dist


def test_sssp_golden(oracle):
    # crates/algos/src/sssp.rs:282-313
    e = [(0, 1, 4.0), (0, 2, 2.0), (1, 2, 5.0), (1, 3, 10.0), (2, 4, 3.0), (3, 5, 11.0), (4, 3, 4.0)]
    s = np.array([x[0] for x in e], np.uint32)
    d = np.array([x[1] for x in e], np.uint32)
    w = np.array([x[2] for x in e], np.float32)
    off, tgt, wv = oracle.csr_build(6, s, d, oracle.OUTGOING, oracle.DEDUPLICATED, w)
    dist = oracle.delta_stepping(off, tgt, wv, 0, 3.0)
    assert np.array_equal(dist, np.array([0, 4, 2, 9, 5, 20], np.float32))
    with pytest.raises(IndexError):
        oracle.delta_stepping(off, tgt, wv, 6, 3.0)


@pytest.mark.parametrize("delta", [0.05, 0.3, 10.0])
def test_sssp_equals_f32_dijkstra(oracle, delta):
    # the result is the least fixed point under f32 round-to-nearest add — schedule-free
    s, d = oracle.rmat_edges(10, seed=7)
    w = oracle.rmat_weights(s.size, seed=44)
    n = 1 << 10
    off, tgt, wv = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED, w)
    start = int(np.flatnonzero(np.diff(off) > 0)[0])
    dist = oracle.delta_stepping(off, tgt, wv, start, delta)
    assert np.array_equal(dist, _dijkstra_f32(off, tgt, wv, start))
    assert (dist == np.finfo(np.float32).max).any()  # unreachable = f32::MAX, not inf


def _min_label_components(n, s, d):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in zip(s.tolist(), d.tolist()):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return np.array([find(x) for x in range(n)], np.uint32)


def test_wcc_relations_and_min_labels(oracle, scale8):
    # crates/algos/src/wcc.rs:307-329
    s, d = _edges([(0, 1), (2, 3)])
    for algo in (oracle.AFFOREST, oracle.AFFOREST_DSS, oracle.BASELINE):
        ooff, otgt = oracle.csr_build(4, s, d, oracle.OUTGOING, oracle.UNSORTED)
        ioff, itgt = oracle.csr_build(4, s, d, oracle.INCOMING, oracle.UNSORTED)
        c = oracle.wcc(ooff, otgt, ioff, itgt, algo)
        assert c[0] == c[1] and c[2] == c[3] and c[1] != c[2]
    # derived: scale_8 -> 16 components, largest 241, 15 isolated; label == min id of the component
    s, d, n = scale8
    ooff, otgt = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    ref = _min_label_components(n, s, d)
    for algo in (oracle.AFFOREST, oracle.AFFOREST_DSS, oracle.BASELINE):
        for rounds in (0, 1, 2, 5):
            for seed in (1, 2, 3):
                c = oracle.wcc(ooff, otgt, ioff, itgt, algo, neighbor_rounds=rounds, seed=seed)
                assert np.array_equal(c, ref)
    assert np.unique(ref).size == 16 and np.bincount(ref).max() == 241
    with pytest.raises(ValueError):
        oracle.wcc(ooff, otgt, ioff, itgt, oracle.AFFOREST, sampling_size=0)


def test_union_find_chains(oracle):
    # crates/algos/src/afforest.rs:121-133, crates/algos/src/dss.rs:182-218
    L = oracle.lib()
    for kind in (0, 1):
        p = np.empty(10, np.uint32)
        L.orc_uf_new(10, p)
        for a, b in [(9, 7), (7, 4), (4, 2), (2, 0)]:
            L.orc_uf_union(kind, p, a, b)
        L.orc_uf_compress(kind, p, 10)
        assert L.orc_uf_find(kind, p, 9) == 0
    p = np.empty(10, np.uint32)
    L.orc_uf_new(10, p)
    L.orc_uf_union(1, p, 2, 4)  # dss.rs doc-test :31-36
    assert L.orc_uf_find(1, p, 2) == 2 and L.orc_uf_find(1, p, 4) == 2


def test_greedy_partition(oracle):
    # crates/builder/src/graph_ops.rs:417-430 doc-test: edges (1,0),(1,2),(2,0),(3,2) ->
    # in_degree_partition(2) == [0..1, 1..4]
    s, d = _edges([(1, 0), (1, 2), (2, 0), (3, 2)])
    ioff, _ = oracle.csr_build(4, s, d, oracle.INCOMING, oracle.UNSORTED)
    assert oracle.greedy_degree_partition(ioff, 2) == [(0, 1), (1, 4)]
    off = np.array([0, 3, 4, 5, 6], np.uint32)  # degrees 3,1,1,1, total 6, batch 3
    assert oracle.greedy_degree_partition(off, 2) == [(0, 1), (1, 4)]
    off = np.array([0, 1, 2, 3, 4], np.uint32)
    assert oracle.greedy_degree_partition(off, 2) == [(0, 2), (2, 4)]
    assert oracle.greedy_degree_partition(off, 8) == [(0, 1), (1, 2), (2, 3), (3, 4)]


def test_rmat_generator_properties(oracle):
    s, d = oracle.rmat_edges(12, seed=42)
    assert s.size == 16 << 12 and s.max() < (1 << 12) and d.max() < (1 << 12)
    s2, d2 = oracle.rmat_edges(12, seed=42, first=1000, count=50)
    assert np.array_equal(s2, s[1000:1050]) and np.array_equal(d2, d[1000:1050])
    # skewed: the top 1% of sources carry a large share of the edges
    deg = np.sort(np.bincount(s, minlength=1 << 12))[::-1]
    assert deg[:41].sum() > 0.2 * s.size
    w = oracle.rmat_weights(1000)
    assert w.min() > 0 and w.max() <= 1


def test_timed_cpu_baseline_runs_the_same_sweeps(oracle):
    """bench.py's cpu_baseline leg (private NUMA-spread copies, timed inside the oracle) does the work it claims:
    after k sweeps its error matches the threaded restatement's (the in-place order is racy beyond one chunk, so
    only to a tolerance), with and without the spread copies; effective_cores() is a positive count."""
    s, d = oracle.rmat_edges(12, seed=4)
    n = 1 << 12
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    _, it, err = oracle.page_rank_chunked(ioff, itgt, od, 6, 0.0, 0.85, threads=1)
    assert it == 6
    for spread in (False, True):
        sec, e = oracle.page_rank_chunked_timed(ioff, itgt, od, 5, 0.85, threads=1, spread=spread)  # 1 untimed + 5 timed
        assert sec > 0.0 and e == err  # one chunk, one thread: the sequential order, bit for bit
    sec, e = oracle.page_rank_chunked_timed(ioff, itgt, od, 5, 0.85, threads=4, spread=True)
    assert abs(e - err) <= 0.2 * err
    assert oracle.effective_cores() >= 1


def test_sssp_reference_stale_check_quirk(oracle):
    """Derived, not reference-published: the reference files a node whose new distance d falls in bin
    (usize)(d/delta) (sssp.rs:192) and later skips it as stale unless d >= delta * bin (sssp.rs:126).  In f32
    13.5 / 0.3 rounds up to 45.0 while 0.3 * 45 = 13.500001, so the node at 13.5 is never relaxed and node 2
    stays at f32::MAX.  The restatement reproduces that; sssp_fixed_point is the intended answer."""
    s, d = np.array([0, 1], np.uint32), np.array([1, 2], np.uint32)
    w = np.array([13.5, 1.0], np.float32)
    off, tgt, wv = oracle.csr_build(3, s, d, oracle.OUTGOING, oracle.SORTED, w)
    fmax = np.finfo(np.float32).max
    assert list(oracle.delta_stepping(off, tgt, wv, 0, 0.3)) == [0.0, 13.5, fmax]
    assert list(oracle.delta_stepping(off, tgt, wv, 0, 0.25)) == [0.0, 13.5, 14.5]
    fp = oracle.sssp_fixed_point(off, tgt, wv, 0)
    assert list(fp) == [0.0, 13.5, 14.5]
    assert list(oracle.stale_check_misfires(fp, 0.3)) == [False, True, False]
    assert not oracle.stale_check_misfires(fp, 0.25).any()
    # wherever the check does not misfire the two agree (random graphs, several deltas)
    rng = np.random.default_rng(3)
    agree = quirky = 0
    for _ in range(60):
        n, m = int(rng.integers(2, 400)), int(rng.integers(1, 3000))
        s, d = rng.integers(0, n, m).astype(np.uint32), rng.integers(0, n, m).astype(np.uint32)
        w = rng.choice(np.array([0.0, 0.125, 0.5, 1.0, 2.75], np.float32), m)
        off, tgt, wv = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED, w)
        for delta in (0.05, 0.3, 3.0):
            start = int(rng.integers(0, n))
            fp = oracle.sssp_fixed_point(off, tgt, wv, start)
            ds = oracle.delta_stepping(off, tgt, wv, start, delta)
            if oracle.stale_check_misfires(fp, delta).any():
                quirky += 1
                assert (ds >= fp).all()  # the reference can only lose improvements
            else:
                agree += 1
                assert np.array_equal(ds, fp)
    assert agree > 100


def test_threaded_timed_legs_return_what_the_sequential_checkers_return(oracle):
    """orc_wcc_afforest_timed / orc_delta_stepping_timed (the reference's rayon threading restated for the TIMED
    cpu_baseline legs, wcc.rs:186-301 / sssp.rs:64-204) against the sequential checkers, several thread counts"""
    O = oracle
    for scale in (10, 14):
        s, d = O.rmat_edges(scale, 42)
        n = 1 << scale
        ooff, otgt = O.csr_build(n, s, d, O.OUTGOING, O.SORTED)
        ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
        ref = O.wcc(ooff, otgt, ioff, itgt)
        w = O.rmat_weights(s.size, 44)
        off, tgt, wv = O.csr_build(n, s, d, O.OUTGOING, O.SORTED, w)
        start = int(np.flatnonzero(np.diff(off) > 0)[0])
        dref = O.delta_stepping(off, tgt, wv, start, 0.1)
        assert not O.stale_check_misfires(dref, 0.1).any() or np.array_equal(dref, O.sssp_fixed_point(off, tgt, wv, start))
        for threads in (1, 2, 5):
            got, secs = O.wcc_afforest_timed(ooff, otgt, ioff, itgt, threads, native=False)
            assert np.array_equal(got, ref) and secs > 0
            dist, secs = O.delta_stepping_timed(off, tgt, wv, start, 0.1, threads, native=False)
            assert np.array_equal(dist.view(np.uint32), dref.view(np.uint32)) and secs > 0
