"""Hub rows of the propagation-blocking PageRank engine follow the reference's left-to-right f32 row sums
(crates/algos/src/page_rank.rs:143-146): on long rows that order has a systematic drift (thousands of equal
terms, each rounded the same way against the running sum), so matching the reference within 1e-5 means
reproducing its rounding.  Since round 4 the hub rows' sums ARE those sums (pb_hubseq_kernel / pb_hublong_kernel).
Checked two ways: one sweep from the same out_scores against the sequential sum (orc_page_rank_jacobi_sweep) — hub
rows bit for bit — and the fixed point against the reference's threaded path on EVERY row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from graph_amd import prelude

    return prelude


def _graph(P, oracle, scale):
    s, d = oracle.rmat_edges(scale, seed=42)
    n = 1 << scale
    out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    return n, P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted), ioff, itgt, oracle.out_degrees_from(n, s)


def _one_sweep(inc, n, od, scores0, x0):
    import torch
    from graph_amd.engine import PageRankEngine

    eng = PageRankEngine(inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    scores = torch.from_numpy(scores0.copy()).cuda()
    x_in = torch.from_numpy(x0.copy()).cuda()
    x_out = torch.empty_like(x_in)
    err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.sweep(x_in, x_out, scores, err)
    torch.cuda.synchronize()
    return scores.cpu().numpy(), float(err.item()), eng.plan_info()


@pytest.mark.parametrize("scale", [20])
def test_one_sweep_hub_rows_match_the_sequential_sum(P, oracle, scale, monkeypatch):
    n, g, ioff, itgt, od = _graph(P, oracle, scale)
    deg = np.diff(ioff.astype(np.int64))
    # realistic inputs: the reference's own fixed point (many sources share one out_score: (1-d)/n / out_degree)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    with np.errstate(divide="ignore"):
        x0 = (ref / od.astype(np.float32)).astype(np.float32)
    x_fin = np.where(np.isfinite(x0), x0, np.float32(0))  # +inf entries (no out-edges) are never gathered
    seq = ref.copy()
    oracle.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, seq, x_fin)  # left-to-right f32 row sums, in place
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    got, _, info = _one_sweep(g.csr_inc, n, od, ref, x0)
    assert info["hub_in_degree"] == 4096 and info["hub_rows"] == int((deg >= 4096).sum()) > 0
    assert info["hub_edges"] == int(deg[deg >= 4096].sum()) and 1 <= info["hub_groups"] <= info["hub_rows"]
    monkeypatch.setenv("GM_PB_HUB_DEG", "0")  # every row exactly rounded: the round-1 behaviour
    exact, _, info0 = _one_sweep(g.csr_inc, n, od, ref, x0)
    assert info0["hub_rows"] == 0 and info0["hub_groups"] == 0
    hub = deg >= 4096
    rel = np.abs(got.astype(np.float64) - seq) / seq
    rel0 = np.abs(exact.astype(np.float64) - seq) / seq
    print(f"scale {scale}: one sweep vs sequential sums, hub rows: emulated order {rel[hub].max():.2e} (rms "
          f"{np.sqrt((rel[hub] ** 2).mean()):.2e}), exactly rounded {rel0[hub].max():.2e}; other rows {rel[~hub].max():.2e}")
    assert np.array_equal(got[~hub], exact[~hub])       # rows below the threshold are untouched
    assert np.array_equal(got[hub], seq[hub])           # hub rows: the reference's own left-to-right f32 sums, bit for bit
    long = max(8192, int(np.sort(deg)[::-1][512]) + 1)   # rows of >= 8192 in-edges, at most 512 of them: pb_hublong_kernel
    assert info["long_rows"] == int((deg >= long).sum()) > 0 and info["long_row_terms"] == int(deg[deg >= long].sum())
    assert rel0[hub].max() > 3e-6                        # what the exactly rounded sum misses at this scale (9e-6)
    assert rel.max() <= 5e-6


@pytest.mark.parametrize("scale", [18, 20])
def test_fixed_point_within_1e5_of_the_reference_on_every_row(P, oracle, scale):
    n, g, ioff, itgt, od = _graph(P, oracle, scale)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    got, it, err = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    again, _, err2 = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    assert np.array_equal(got, again) and err == err2   # deterministic
    rel = np.abs(got.astype(np.float64) - ref) / ref
    deg = np.diff(ioff.astype(np.int64))
    print(f"scale {scale}: {it} sweeps, max rel vs the reference on every row {rel.max():.2e} "
          f"(rows with >= 4096 in-edges: {rel[deg >= 4096].max() if (deg >= 4096).any() else 0:.2e})")
    assert rel.max() <= 1e-5


@pytest.mark.parametrize("hub_long", ["4096", "20000", "1000000000"])
def test_hub_rows_give_the_same_bits_whichever_kernel_sums_them(P, oracle, monkeypatch, hub_long):
    """Rows of at least GM_PB_HUB_LONG (32768) in-edges are summed by pb_hublong_kernel (runs of 16 terms composed by a scan),
    shorter hub rows by pb_hubseq_kernel (one lane per row): both compute the reference's left-to-right f32 sum, so the
    threshold between them must not change a bit — every hub row through the scan, the usual split, every one through the
    lane walk."""
    n, g, ioff, itgt, od = _graph(P, oracle, 20)
    cfg = P.PageRankConfig(30, 0.0, 0.85)
    base, _, eb = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    monkeypatch.setenv("GM_PB_HUB_LONG", hub_long)
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    got, _, eg = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    assert np.array_equal(got, base) and eg == eb


def test_hub_rows_take_their_hot_terms_off_the_value_stream_without_changing_a_bit(P, oracle, monkeypatch):
    """Terms of hub rows that come from hot sources do not pass the value stream (12 B per edge) but are gathered from the hot
    sources' table by pb_hubseq_kernel at their place in the row's order (4-byte records; GM_PB_HUB_HOT=0: all through the
    stream).  Same sums, bit for bit; and the plan says how many edges took the short way."""
    from graph_amd.engine import PageRankEngine
    import torch

    n, g, ioff, itgt, od = _graph(P, oracle, 20)
    cfg = P.PageRankConfig(25, 0.0, 0.85)
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    with_hot, _, e1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    eng = PageRankEngine(g.csr_inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    info = eng.plan_info()
    del eng
    monkeypatch.setenv("GM_PB_HUB_HOT", "0")
    without, _, e0 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    eng = PageRankEngine(g.csr_inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    info0 = eng.plan_info()
    del eng
    print(f"scale 20: {info['hub_hot_edges']} of {info['hub_edges']} hub edges off the stream; value entries {info0['value_entries']} -> {info['value_entries']}")
    assert info["hub_hot_edges"] > 0 and info0["hub_hot_edges"] == 0
    assert info0["value_entries"] - info["value_entries"] >= info["hub_hot_edges"]  # (+ the padding that went with them)
    assert np.array_equal(with_hot, without) and e1 == e0


def test_rows_of_equal_terms_below_the_threshold_follow_the_reference(P, oracle, monkeypatch):
    """Rows BELOW the hub threshold are exactly rounded sums.  When a row's terms are EQUAL — a node with k leaf followers
    (in-degree 0, out-degree 1: each at exactly (1 - d) / n) — the reference's left-to-right f32 sum (page_rank.rs:143-146)
    drifts systematically and an exact sum does not follow it: by an amount that depends on n through the bits of (1 - d) / n and
    exceeds the north-star 1e-5 for most n once k >= 2000 (the model below: a cumsum in f32 against the rounded product; found in
    round 6, profiles/r06_leaf_fan_probe.txt).  Since then a row with at least 512 sources that have no in-edges themselves
    (GM_PB_HUB_LEAVES) is a hub row whatever its length — summed the reference's way: its bits.  Checked here at an n where the drift
    is large (6e-5 at 4095 terms): the default plan against the reference on every row, the same graph cut over three ranks (the
    partitioned front hands its slices the flags: the single engine's bits), and — with the rule off — that the deviation is what the
    model says (so the test would notice the rule going missing)."""
    scale, fans = 16, [300, 511, 1000, 2687, 4095]
    s, d = oracle.rmat_edges(scale, seed=42)
    n0 = 1 << scale
    centres = n0 + np.arange(len(fans))
    at, ls, ld = n0 + len(fans), [], []
    for c, k in zip(centres, fans):
        ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
    # ... and the other way equal terms arise: the SIBLINGS of one parent (a front page and the 3000 pages only it links to, which all
    # link to one other node): one in-edge each, from the same node, so one score each — 3000 equal terms on the node they point at
    parent, target, kids = at, at + 1, 3000
    at += 2
    sib = np.arange(at, at + kids, dtype=np.uint32); at += kids
    ls += [np.full(kids, parent, np.uint32), sib, np.array([1, target], np.uint32)]
    ld += [sib, np.full(kids, target, np.uint32), np.array([parent, 0], np.uint32)]  # (node 1 -> parent, target -> node 0)
    s = np.concatenate([s] + ls + [centres.astype(np.uint32)])
    d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])  # (every centre points at node 0: no sink)
    n = 297676  # isolated nodes up to an n at which the drift is large
    assert int(at) <= n
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    monkeypatch.setenv("GM_MULTI_ENGINE", "pb")

    def run(devices=None):
        out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
        inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
        g = P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted)
        cfg = P.PageRankConfig(200, 1e-10, 0.85)
        got, _, _ = P.page_rank(g, cfg, P.PageRankMode.JacobiPB) if devices is None else P.page_rank_multi(g, cfg, devices=devices)
        return np.asarray(got)

    got = run()
    assert np.array_equal(run(devices=[0, 0, 0]), got)          # three virtual ranks: the same bits
    rel = np.abs(got.astype(np.float64) - ref) / ref
    print(f"n {n}, fans of {fans} equal terms, default plan: max rel on every row {rel.max():.2e}; the fans' rows {rel[centres]}")
    assert np.array_equal(got[centres[2:]], ref[centres[2:]])  # 1000 / 2687 / 4095 leaf sources: summed the reference's way, its bits
    print(f"   the node 3000 siblings point at: {rel[target]:.2e}")
    assert rel[target] <= 2e-6                                  # (its terms are the reference's up to what the siblings' scores differ by)
    assert rel.max() <= 1e-5                                    # (300 and 511 stay exactly rounded sums: within the tolerance)
    monkeypatch.setenv("GM_PB_HUB_LEAVES", "0")                 # the rule off: what the plans did until round 6
    off = run()
    v = np.float32((np.float32(1.0) - np.float32(0.85)) / np.float32(n))  # a leaf's score = its out_score (out-degree 1)
    for c, k in zip(centres, fans):
        seq = np.cumsum(np.full(k, v, np.float32), dtype=np.float32)[-1]       # the reference's sum
        exact = np.float32(float(v) * k)                                      # the exactly rounded one
        model = abs(float(seq) - float(exact)) * 0.85 / float(ref[c])
        r = abs(float(off[c]) - float(ref[c])) / float(ref[c])
        print(f"   rule off, fan of {k} equal terms: device vs reference {r:.2e}, the model's drift {model:.2e}")
        assert abs(r - model) <= 0.1 * model + 2e-7
    rel_off = np.abs(off.astype(np.float64) - ref) / ref
    print(f"   rule off: {int((rel_off > 1e-5).sum())} rows over 1e-5 (max {rel_off.max():.2e}): the fans' rows and what is downstream of them; "
          f"the siblings' node {rel_off[target]:.2e}")
    assert rel_off[centres[-1]] > 1e-5 and rel_off.max() <= 1e-4


def test_python_front_slices_with_source_flags_give_the_single_engines_bits(P, oracle):
    """The one-process-per-rank front (graph_amd/distributed.py) on the graph of the test above: three virtual ranks, a sweep in two
    pieces each, the slices' handles given the per-slot flags (distributed.source_flags -> DeviceCsr.set_source_flags) — the single
    engine's bits after every sweep; WITHOUT the flags the slices cannot see that the fans' sources have no in-edges, sum those rows
    exactly rounded, and differ from the single engine (which is why the flags exist)."""
    import ctypes as C

    import torch
    from graph_amd._lib import check, lib, vp
    from graph_amd.distributed import PiecewiseExchange, greedy_degree_partition, pad_bounds, source_flags, split_exchange_layout
    from graph_amd.engine import PageRankEngine

    scale, fans, world, sweeps = 16, [1000, 2687, 4095], 3, 6
    s, d = oracle.rmat_edges(scale, seed=42)
    n0 = 1 << scale
    centres = n0 + np.arange(len(fans))
    at, ls, ld = n0 + len(fans), [], []
    for c, k in zip(centres, fans):
        ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
    s = np.concatenate([s] + ls + [centres.astype(np.uint32)])
    d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])
    n = 297676
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    ioff = inc.host()[0]
    dev = torch.device("cuda", 0)
    od = torch.from_numpy(oracle.out_degrees_from(n, s).astype(np.int32)).cuda()
    no_in = torch.from_numpy((np.diff(ioff.astype(np.int64)) <= 1).astype(np.uint8)).cuda()  # at most one in-edge
    bounds, _ = pad_bounds(greedy_degree_partition(ioff, world), world, n)
    eng = PageRankEngine(inc.handle, n, 0, od, 0.85, engine=PageRankEngine.PB)
    assert eng.plan_info()["hub_rows"] >= 3  # (the three fans among them: flagged by the whole graph's plan from its own offsets)
    x = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
    sc = torch.zeros(n, device=dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    eng.init(sc, x[0])
    for k in range(sweeps):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    lay = split_exchange_layout(od, bounds, parts=2)
    results = {}
    for with_flags in (True, False):
        shared = [torch.zeros(lay["x_len"], device=dev) for _ in range(2)]
        ranks = []
        for r in range(world):
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            h = vp()
            check(lib().gm_csr_slice_rows_map(inc.handle, lo, hi, lay["node_map"].data_ptr(), C.byref(h)))
            csr = P.DeviceCsr(h)
            if with_flags:
                csr.set_source_flags(source_flags(lay["node_map"], no_in, lay["x_len"]))
            e = PageRankEngine(csr.handle, n, lo, od[lo:hi].contiguous(), 0.85, x_len=lay["x_len"], engine=PageRankEngine.PB)

            def gather(dst_views, src, k, r=r):  # this rank's slot of region k
                dst_views[r].copy_(src)

            ex = PiecewiseExchange(e, lay, r, hi - lo, dev, gather=gather, x=shared)
            ranks.append((csr, e, ex, torch.zeros(hi - lo, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)))
        for (_, _, ex, scl, _) in ranks:
            ex.start(scl)
        for k in range(sweeps):
            for (_, _, ex, scl, el) in ranks:
                ex.sweep(scl, el)
        results[with_flags] = torch.cat([rk[3] for rk in ranks])
    assert torch.equal(results[True], sc)
    assert not torch.equal(results[False], sc)
    worst = float(((results[False] - sc).abs() / sc).max())
    print(f"three virtual ranks, {sweeps} sweeps: with the flags the single engine's bits; without them max rel {worst:.2e} from it")


def test_small_graph_with_rows_of_equal_terms_gets_the_engine_that_follows_the_reference(P, oracle, monkeypatch):
    """Below 2^24 edges the default call runs on the pull tiles, which sum a long row as a tree — unless some row must be summed in
    the reference's order: a row of >= 4096 entries, or (round 6) a shorter one with >= 512 sources of at most one in-edge, i.e. many
    equal terms.  A graph of 300 K edges with two leaf fans and no long row: the default call gives the fans' rows the reference's
    bits; with the rule off it takes the pull tiles and the 3000-fan is off by the reference's drift."""
    nb, fans = 100_000, [2000, 3000]
    base = np.arange(nb, dtype=np.uint32)
    s = np.concatenate([base, base])
    d = np.concatenate([(base * 7 + 3) % nb, (base * 13 + 5) % nb]).astype(np.uint32)  # in-degree 2 everywhere
    centres = nb + np.arange(len(fans))
    at, ls, ld = nb + len(fans), [], []
    for c, k in zip(centres, fans):
        ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
    s = np.concatenate([s] + ls + [centres.astype(np.uint32)])
    d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])
    n = 297676
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    assert int(np.diff(ioff.astype(np.int64)).max()) == 3000
    od = oracle.out_degrees_from(n, s)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)

    def run():
        out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
        inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
        got, _, _ = P.page_rank(P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted), P.PageRankConfig(200, 1e-10, 0.85))
        return np.asarray(got)

    got = run()
    rel = np.abs(got.astype(np.float64) - ref) / ref
    print(f"default call, {s.size} edges, fans of {fans}: max rel on every row {rel.max():.2e}, the fans' rows {rel[centres]}")
    assert np.array_equal(got[centres], ref[centres]) and rel.max() <= 1e-5
    monkeypatch.setenv("GM_PB_HUB_LEAVES", "0")
    off = run()
    rel_off = np.abs(off.astype(np.float64) - ref) / ref
    print(f"   rule off (pull tiles): the fans' rows {rel_off[centres]}")
    assert rel_off[centres[1]] > 1e-5
