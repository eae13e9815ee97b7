"""The hub-row path of the propagation-blocking PageRank engine (rows with >= 4096 in-edges are summed in the reference's
left-to-right f32 order, crates/algos/src/page_rank.rs:143-146) on inputs that are NOT RMAT seed 42: term sequences built
to stress the emulation (ascending / descending / alternating magnitudes, running sums that end at a power of two, many
rows in one group), other RMAT seeds, CsrLayout::Unsorted, and random small graphs with the hub threshold lowered so
that ordinary rows take the hub path.  One sweep is compared against orc_page_rank_jacobi_sweep (the same sweep with
sequential f32 row sums) — since round 4 for EQUALITY on the hub rows, which are summed the reference's way — fixed points
against orc_page_rank_chunked; the bar is north_star's 1e-5 on every row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from graph_amd import prelude

    return prelude


def _sweep(P, n, src, dst, x0, scores0, layout=None):
    """one synchronous sweep of the PB engine from (scores0, x0) and the same sweep with sequential row sums"""
    import torch
    from graph_amd.engine import PageRankEngine
    from oracle import oracle as O

    layout = P.CsrLayout.Sorted if layout is None else layout
    inc = P.DeviceCsr.from_edges(n, src, dst, None, P.Direction.Incoming, layout)
    ioff, itgt, _ = inc.host()
    od = np.bincount(src, minlength=n).astype(np.uint32)
    eng = PageRankEngine(inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    scores = torch.from_numpy(scores0.copy()).cuda()
    x_in = torch.from_numpy(x0.copy()).cuda()
    x_out = torch.empty_like(x_in)
    err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.sweep(x_in, x_out, scores, err)
    torch.cuda.synchronize()
    info = eng.plan_info()
    seq = scores0.copy()
    O.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, seq, np.where(np.isfinite(x0), x0, np.float32(0)))
    return scores.cpu().numpy(), seq, info, np.diff(ioff.astype(np.int64))


def _star(hubs, sources):
    """every one of `sources` nodes points to every one of `hubs` nodes (ids 0 .. hubs-1); the sources follow"""
    s = np.repeat(np.arange(hubs, hubs + sources, dtype=np.uint32), hubs)
    d = np.tile(np.arange(hubs, dtype=np.uint32), sources)
    return hubs + sources, s, d


TERMS = {
    "ascending": lambda k, rng: np.geomspace(1e-13, 1e-7, k),
    "descending": lambda k, rng: np.geomspace(1e-7, 1e-13, k),
    "alternating 2^+-12": lambda k, rng: np.where(np.arange(k) % 2 == 0, 2.0 ** -24, 2.0 ** -36),
    "blocks of 2^+-12": lambda k, rng: np.where((np.arange(k) // 5000) % 2 == 0, 2.0 ** -36, 2.0 ** -24),
    "lognormal": lambda k, rng: np.exp(rng.normal(-20.0, 3.0, k)),
    "equal, sum ends on a power of two": lambda k, rng: np.full(k, 2.0 ** -30),
    "equal, odd mantissa": lambda k, rng: np.full(k, np.float32(1.2345678e-9)),
    "one giant first": lambda k, rng: np.concatenate([[1e-3], np.full(k - 1, 3e-10)]),
    "one giant last": lambda k, rng: np.concatenate([np.full(k - 1, 3e-10), [1e-3]]),
}


@pytest.mark.parametrize("hubs,sources", [(1, 1 << 20), (1, (1 << 20) + 1), (2, 300_000), (5, 1 << 17), (40, 9000), (64, 4500),
                                          (2, 5000), (3, 7001), (4, 4500)])  # (the last three with GM_PB_HUB_THIN=4: groups of few rows, row by row through the long-row kernel)
@pytest.mark.parametrize("kind", list(TERMS))
def test_one_sweep_of_adversarial_term_sequences_matches_the_sequential_sum(P, monkeypatch, hubs, sources, kind):
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    if hubs <= 4 and sources < 8192:
        monkeypatch.setenv("GM_PB_HUB_THIN", "4")
    rng = np.random.default_rng(hubs * 1000003 + sources)
    n, s, d = _star(hubs, sources)
    x0 = np.full(n, np.inf, np.float32)  # hubs have no out-edges: never gathered
    x0[hubs:] = TERMS[kind](sources, rng).astype(np.float32)
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    got, seq, info, deg = _sweep(P, n, s, d, x0, scores0)
    assert info["hub_rows"] == hubs and info["hub_edges"] == hubs * sources
    assert np.array_equal(got[hubs:], seq[hubs:])  # rows without in-edges: base score, bit for bit
    # the hub rows: the reference's left-to-right f32 sum of the same terms, bit for bit (rounds 2-3: an emulation within
    # 3e-6, 6.4e-6 with one giant term last)
    assert np.array_equal(got[:hubs], seq[:hubs]), np.abs(got[:hubs].astype(np.float64) - seq[:hubs]).max()


def test_item_boundary_sums_that_hop_over_a_power_of_two_between_sweeps(P, monkeypatch):
    """A long row is summed by several workgroups, each forming the pairs of its passes AHEAD on the binade the sums at the
    pass boundaries had in the sweep before (pb_hublong_kernel).  Two input vectors are swept in turn whose boundary sums lie
    on either side of a power of two — every prediction is wrong in every sweep, and the neighbouring items overwrite the
    boundary sums this item predicts from while it runs: whatever a workgroup reads, the row's sum must be the sequential one,
    sweep after sweep (the first version let every thread read the boundary sums for itself; threads that saw different
    exponents disagreed about the barriers they would meet)."""
    import torch
    from graph_amd.engine import PageRankEngine
    from oracle import oracle as O

    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    monkeypatch.setenv("GM_PB_LONG_PASSES", "2")
    hubs, sources = 64, 1 << 19  # 64 rows of 32 items of 16384 terms: 2048 workgroups, most of which start while others finish
    n, s, d = _star(hubs, sources)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    ioff, itgt, _ = inc.host()
    od = np.bincount(s, minlength=n).astype(np.uint32)
    eng = PageRankEngine(inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    assert eng.plan_info()["long_rows"] == hubs
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    xs, want = [], []
    for c in (0.97, 1.03):  # 16384 terms of c * 2^-40: the first item ends just below / just above 2^-26, the second 2^-25, ...
        x0 = np.full(n, np.inf, np.float32)
        x0[hubs:] = np.float32(c) * np.float32(2.0 ** -40)
        seq = scores0.copy()
        O.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, seq, np.where(np.isfinite(x0), x0, np.float32(0)))
        xs.append(torch.from_numpy(x0).cuda())
        want.append(seq[:hubs].copy())
    assert not np.array_equal(want[0], want[1])
    scores = torch.from_numpy(scores0.copy()).cuda()
    x_out = torch.empty_like(xs[0])
    err = torch.zeros(1, dtype=torch.float64, device="cuda")
    for k in range(60):
        eng.sweep(xs[k % 2], x_out, scores, err)
        got = scores[:hubs].cpu().numpy()
        assert np.array_equal(got, want[k % 2]), (k, got, want[k % 2])
    for k in range(20):  # ... and with predictions that hold (the same vector again and again: the pairs formed ahead are used)
        eng.sweep(xs[0], x_out, scores, err)
        assert np.array_equal(scores[:hubs].cpu().numpy(), want[0]), k


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_other_rmat_seeds_fixed_point_within_1e5_on_every_row(P, oracle, seed):
    scale, n = 18, 1 << 18
    s, d = oracle.rmat_edges(scale, seed=seed)
    g = P.DirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted),
                           P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    got, it, _ = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    rel = np.abs(got.astype(np.float64) - ref) / ref
    deg = np.diff(ioff.astype(np.int64))
    print(f"RMAT scale {scale} seed {seed}: {it} sweeps, max rel {rel.max():.2e} on every row, {int((deg >= 4096).sum())} hub rows "
          f"(max in-degree {int(deg.max())}): {rel[deg >= 4096].max() if (deg >= 4096).any() else 0:.2e}")
    assert rel.max() <= 1e-5, rel.max()


def test_unsorted_layout_one_sweep_and_fixed_point(P, oracle, monkeypatch):
    """CsrLayout::Unsorted (the reference's DEFAULT layout, csr.rs:34-45): a row's in-neighbours lie in arrival order in the
    CSR, and that is the order the reference adds them in (page_rank.rs:143-146).  The plan notices that the hub rows' lists
    are not ascending and sums them through a per-term index IN CSR ORDER (pb_hublong_kernel<true>) instead of in the order
    the value stream delivers them in (ascending source).  Against the oracle's build of the same layout — the same arrival
    order on both sides — the fixed point meets the bar on EVERY row (round 4: 1.05e-5 on the hub rows, bar 2.5e-5)."""
    scale, n = 18, 1 << 18
    s, d = oracle.rmat_edges(scale, seed=7)
    g = P.DirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Unsorted),
                           P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Unsorted), P.CsrLayout.Unsorted)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.UNSORTED)
    got_off, got_tgt, _ = g.csr_inc.host()
    assert np.array_equal(got_off, ioff) and np.array_equal(got_tgt, itgt)  # same arrival order on both sides
    od = oracle.out_degrees_from(n, s)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    got, it, _ = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    rel = np.abs(got.astype(np.float64) - ref) / ref
    deg = np.diff(ioff.astype(np.int64))
    assert int((deg >= 4096).sum()) >= 10
    print(f"Unsorted layout, scale {scale}: {it} sweeps, max rel {rel.max():.2e} (hub rows {rel[deg >= 4096].max():.2e}, "
          f"other rows {rel[deg < 4096].max():.2e})")
    assert rel.max() <= 1e-5, rel.max()
    assert rel[deg >= 4096].max() <= 2e-6  # the hub rows' sums are the reference's own: what is left is what their inputs differ by
    # in source order all the same (GM_PB_HUB_CSR=0, round 4's behaviour): a valid f32 sum of the same terms, not the reference's
    monkeypatch.setenv("GM_PB_HUB_CSR", "0")
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    old, _, _ = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    rel_old = np.abs(old.astype(np.float64) - ref) / ref
    assert rel_old[deg >= 4096].max() > rel[deg >= 4096].max()


@pytest.mark.parametrize("hubs,sources", [(1, (1 << 18) + 3), (3, 70_001), (40, 9000), (64, 4500)])
@pytest.mark.parametrize("kind", ["lognormal", "alternating 2^+-12", "one giant first"])
def test_one_sweep_of_shuffled_lists_matches_the_sequential_sum_in_csr_order(P, monkeypatch, hubs, sources, kind):
    """Hub rows whose lists are NOT ascending: one sweep from given out_scores reproduces orc_page_rank_jacobi_sweep — the
    left-to-right f32 sum over the list AS IT LIES IN THE CSR — bit for bit, through every length class of
    pb_hublong_kernel<true> (one pass, several passes, several items of one row)."""
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    rng = np.random.default_rng(hubs * 7919 + sources)
    n, s, d = _star(hubs, sources)
    perm = rng.permutation(s.size)  # the edges in a random arrival order: CsrLayout::Unsorted keeps it inside every list
    s, d = s[perm], d[perm]
    x0 = np.full(n, np.inf, np.float32)
    x0[hubs:] = TERMS[kind](sources, rng).astype(np.float32)
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    got, seq, info, deg = _sweep(P, n, s, d, x0, scores0, layout=P.CsrLayout.Unsorted)
    assert info["hub_rows"] == hubs and info["hub_edges"] == hubs * sources
    assert np.array_equal(got, seq), np.abs(got[:hubs].astype(np.float64) - seq[:hubs]).max()


def test_random_small_graphs_through_the_hub_path(P, monkeypatch):
    """tools/fuzz_parity.py pins GM_PB_HUB_DEG=0; here the threshold is 64, so the rows of random graphs with planted hubs
    take the hub path (groups of many short rows, first steps without a binade, rows of a few dozen terms)."""
    from oracle import oracle as O

    monkeypatch.setenv("GM_PB_HUB_DEG", "64")
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    rng = np.random.default_rng(20260925)
    worst = worst4 = 0.0
    for case in range(40):
        n = int(rng.integers(200, 6000))
        m = int(rng.integers(2000, 60000))
        s = rng.integers(0, n, m).astype(np.uint32)
        d = rng.integers(0, n, m).astype(np.uint32)
        for _ in range(int(rng.integers(1, 6))):  # planted hubs: a share of the edges points at one node
            a, b = sorted(rng.integers(0, m, 2))
            d[a:b] = rng.integers(0, n)
        inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
        out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
        g = P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted)
        ioff, itgt, _ = inc.host()
        od = np.bincount(s, minlength=n).astype(np.uint32)
        worst_of = {}
        for sweeps in (1, 4):
            got, it, _ = P.page_rank(g, P.PageRankConfig(sweeps, 0.0, 0.85), P.PageRankMode.JacobiPB)
            assert it == sweeps
            # the same synchronous sweeps with sequential f32 row sums
            scores = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
            with np.errstate(divide="ignore"):
                x = (scores / od.astype(np.float32)).astype(np.float32)
            for _ in range(sweeps):
                x_fin = np.where(np.isfinite(x), x, np.float32(0))
                x, _ = O.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, scores, x_fin)
            worst_of[sweeps] = float((np.abs(got.astype(np.float64) - scores) / scores).max())
        worst = max(worst, worst_of[1])
        worst4 = max(worst4, worst_of[4])
        # every row of these graphs with >= 64 in-edges takes the hub path: one sweep equals the sequential sums bit for bit
        # on those rows; the other rows are exactly rounded sums (<= 64 terms: within an ulp or two of the sequential sum),
        # and later sweeps add what the graph's feedback makes of that
        assert worst_of[1] <= 3e-6, (case, n, m, worst_of)
        assert worst_of[4] <= 6e-6, (case, n, m, worst_of)
    print(f"40 random graphs through the hub path (threshold 64): worst row after one sweep {worst:.2e} from the sequential sums, "
          f"after four sweeps {worst4:.2e}")


@pytest.mark.parametrize("hubs,sources,kind", [(1, 1 << 20, "lognormal"), (1, (1 << 20) + 1, "ascending"), (2, 300_000, "descending"),
                                               (2, 300_000, "one giant last"), (1, 1 << 20, "one giant first"), (2, 5000, "lognormal"),
                                               (3, 70_000, "alternating 2^+-12"), (1, 40_000, "equal, sum ends on a power of two")])
@pytest.mark.parametrize("hub_long", ["4096", "1000000000"])
def test_long_rows_summed_by_the_scan_and_by_the_lane_walk_give_the_sequential_sum(P, monkeypatch, hubs, sources, kind, hub_long):
    """pb_hublong_kernel (GM_PB_HUB_LONG=4096: every hub row; runs of 16 terms as (count from even J, count from odd J) pairs,
    composed by a scan, the run that leaves the binade added term by term) and pb_hubseq_kernel (threshold out of reach:
    one lane per row, one v_add_f32 per term): the bits of the sequential f32 sum, page_rank.rs:143-146."""
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    monkeypatch.setenv("GM_PB_HUB_LONG", hub_long)
    rng = np.random.default_rng(hubs * 7919 + sources)
    n, s, d = _star(hubs, sources)
    x0 = np.full(n, np.inf, np.float32)
    x0[hubs:] = TERMS[kind](sources, rng).astype(np.float32)
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    got, seq, info, _ = _sweep(P, n, s, d, x0, scores0)
    assert info["hub_rows"] == hubs and info["long_rows"] == (hubs if hub_long == "4096" else 0)
    assert np.array_equal(got, seq)


def test_ties_to_even_inside_the_scan(P, monkeypatch):
    """Terms that end in exactly half an ulp of the running sum (ties) are what makes a run's count depend on the parity
    of the sum's last bit: a row whose terms are all 1.5 ulps of a sum near 2^-10, then all 0.5 ulps, then mixed."""
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    monkeypatch.setenv("GM_PB_HUB_LONG", "4096")
    hubs, sources = 2, 50_000
    n, s, d = _star(hubs, sources)
    u = np.float32(2.0 ** -33)  # ulp of sums in [2^-10, 2^-9)
    terms = np.empty(sources, np.float32)
    terms[0] = np.float32(2.0 ** -10)
    terms[1:20000] = np.float32(1.5) * u
    terms[20000:35000] = np.float32(0.5) * u
    terms[35000:] = np.where(np.arange(sources - 35000) % 3 == 0, np.float32(2.5) * u, np.float32(0.75) * u)
    x0 = np.full(n, np.inf, np.float32)
    x0[hubs:] = terms
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    got, seq, info, _ = _sweep(P, n, s, d, x0, scores0)
    assert info["long_rows"] == hubs
    assert np.array_equal(got, seq)
