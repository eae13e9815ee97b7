"""gm_page_rank_multi: the 1-D partitioned page_rank behind the C ABI (graph_amd/csrc/multi.hip) — partition,
index rewrite, sweep drivers and the exchange, with RCCL on the devices this box has (one: a communicator of
size 1) and with virtual ranks (several ranks on device 0, copies standing in for the all-gather)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from graph_amd import prelude

    return prelude


def _graph(P, oracle, scale, seed=42):
    s, d = oracle.rmat_edges(scale, seed=seed)
    n = 1 << scale
    out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    return P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted), np.bincount(d, minlength=n)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_partitioned_page_rank_matches_the_single_gpu_engine(P, oracle, devices, monkeypatch):
    monkeypatch.setenv("GM_MULTI_ENGINE", "pb")  # the same engine on both sides: exactly rounded row sums
    g, indeg = _graph(P, oracle, 17)
    cfg = P.PageRankConfig(7, 0.0, 0.85)
    one, it1, err1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    got, it, err = P.page_rank_multi(g, cfg, devices=devices)
    assert it == it1 == 7
    # ordinary rows are exactly rounded sums, hub rows (>= 4096 in-edges) the reference's own left-to-right f32 sums
    # (page_rank.rs:143-146) computed bit for bit: neither depends on how the rows are cut over the ranks
    assert int((indeg >= 4096).sum()) > 0
    assert np.array_equal(got, one)
    assert abs(err - err1) <= 1e-5 * max(err1, 1e-30)  # (the ranks' error shares are added in another order)
    again, _, err2 = P.page_rank_multi(g, cfg, devices=devices)
    assert np.array_equal(got, again) and err == err2  # deterministic
    # with every row exactly rounded (GM_PB_HUB_DEG=0) the row sums do not depend on the partition at all
    monkeypatch.setenv("GM_PB_HUB_DEG", "0")
    monkeypatch.setenv("GM_PB_NOCACHE", "1")
    one0, _, _ = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    got0, _, _ = P.page_rank_multi(g, cfg, devices=devices)
    assert np.array_equal(got0, one0)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_partitioned_page_rank_matches_the_reference_directly(P, oracle, devices):
    """The partitioned engine against the ORACLE (the reference's threaded path, page_rank.rs:113-168), both at their
    fixed points, on every row — not through the single-GPU engine.  Scale 18: hub rows (>= 4096 in-edges) exist and
    are cut by the rank boundaries (in-degree ranges, graph_ops.rs:431-439,479-509)."""
    g, indeg = _graph(P, oracle, 18)
    assert int((indeg >= 4096).sum()) > 0
    cfg = P.PageRankConfig(200, 1e-10, 0.85)
    got, it, _ = P.page_rank_multi(g, cfg, devices=devices)
    ioff, itgt, _ = g.csr_inc.host()
    ref, it_ref, _ = oracle.page_rank_chunked(ioff, itgt, g.csr_out.degrees().astype(np.uint32), 200, 1e-10, 0.85)
    rel = np.abs(got.astype(np.float64) - ref) / ref
    print(f"{len(devices)} virtual ranks, scale 18: {it} sweeps (reference {it_ref}); max rel {rel.max():.2e} on every row, "
          f"{rel[indeg >= 4096].max():.2e} on hub rows")
    assert rel.max() <= 1e-5, rel.max()
    assert rel.max() <= 8e-6


def test_partitioned_page_rank_stop_rule_and_default_engines(P, oracle):
    g, _ = _graph(P, oracle, 15, seed=7)
    ref, it_ref, err_ref = P.page_rank(g, P.PageRankConfig(), P.PageRankMode.Jacobi)
    for devices in ([0], [0, 0, 0]):
        got, it, err = P.page_rank_multi(g, P.PageRankConfig(), devices=devices)
        assert it == it_ref and (err < 1e-4 or it == 20)
        np.testing.assert_allclose(got, ref, rtol=4e-6, atol=0)
    # n_devices beyond what the box shows is an error, not a crash
    from graph_amd._lib import GraphMI355XError
    import graph_amd

    with pytest.raises(GraphMI355XError):
        P.page_rank_multi(g, P.PageRankConfig(), n_devices=graph_amd.device_count() + 1)
    # max_iterations = 1: one sweep
    got, it, _ = P.page_rank_multi(g, P.PageRankConfig(1, 1e-4, 0.85), devices=[0, 0])
    assert it == 1


def test_partitioned_page_rank_more_ranks_than_rows_with_edges(P):
    # 6 nodes, 3 virtual ranks, empty trailing ranges
    g = P.GraphBuilder().csr_layout(P.CsrLayout.Sorted).edges([(0, 1), (1, 2), (0, 2), (3, 4), (4, 5), (3, 5)]).build(P.DirectedCsrGraph)
    ref = P.page_rank(g, P.PageRankConfig(5, 0.0, 0.85), P.PageRankMode.Jacobi)
    got = P.page_rank_multi(g, P.PageRankConfig(5, 0.0, 0.85), devices=[0, 0, 0])
    np.testing.assert_allclose(got[0], ref[0], rtol=2e-7, atol=0)
    assert got[1] == 5


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_overlapped_regions_give_the_bits_of_the_blocking_exchange_and_reuse_the_resident_state(P, oracle, devices, monkeypatch, capfd):
    """The exchange in K = 2 regions that travel under the work (the default), the one-region path (GM_MULTI_PARTS=1)
    and the single-GPU engine: with exactly rounded rows the same bits.  The run's partition / slices / engines /
    streams are parked in the in-CSR handle: a second call on the same device list builds nothing."""
    import time

    monkeypatch.setenv("GM_MULTI_ENGINE", "pb")
    monkeypatch.setenv("GM_PB_HUB_DEG", "0")
    monkeypatch.setenv("GM_LOG", "1")
    g, _ = _graph(P, oracle, 18, seed=3)
    cfg = P.PageRankConfig(9, 0.0, 0.85)
    one, _, err1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    capfd.readouterr()
    t0 = time.perf_counter()
    two, it, err = P.page_rank_multi(g, cfg, devices=devices)       # builds the resident state, two regions
    t_first = time.perf_counter() - t0
    log_first = capfd.readouterr().err
    t0 = time.perf_counter()
    again, _, err_again = P.page_rank_multi(g, cfg, devices=devices)  # ... and runs on it again
    t_again = time.perf_counter() - t0
    log_again = capfd.readouterr().err
    assert it == 9 and np.array_equal(two, one) and np.array_equal(again, one) and err == err_again
    assert "2 region(s) overlapped with the work" in log_first and "0 host synchronisation(s)" not in log_first  # the last sweep's error is read
    assert "1 host synchronisation(s)" in log_first               # tolerance 0: only the final error read-back
    assert "multi: row slices + engines" in log_first and "multi: row slices + engines" not in log_again
    print(f"{len(devices)} virtual ranks: first call {t_first * 1e3:.1f} ms, second call on the resident state {t_again * 1e3:.1f} ms")
    monkeypatch.setenv("GM_MULTI_PARTS", "1")
    blocking, _, err_b = P.page_rank_multi(g, cfg, devices=devices)   # another layout: rebuilt, one region
    log_b = capfd.readouterr().err
    assert np.array_equal(blocking, one) and "1 region(s)" in log_b and "multi: row slices + engines" in log_b
    # a tolerance that stops the run early: the stop rule reads the error every sweep, results unchanged
    monkeypatch.delenv("GM_MULTI_PARTS")
    ref, it_ref, _ = P.page_rank(g, P.PageRankConfig(50, 1e-7, 0.85), P.PageRankMode.JacobiPB)
    got, it_got, _ = P.page_rank_multi(g, P.PageRankConfig(50, 1e-7, 0.85), devices=devices)
    assert it_got == it_ref and np.array_equal(got, ref)


@pytest.mark.parametrize("ranks", [2, 5])
def test_pieces_built_without_the_whole_graph_give_the_bits_of_the_sliced_whole(P, ranks, monkeypatch):
    """gm_page_rank_multi_slices: every rank's rows are built from the edges whose destination lies in its range (a chunked
    scan of the counter-based R-MAT generator: no device ever holds the edge list or the whole CSR), the exchange layout is
    derived per rank from the out-degree vector.  Same partitioner, same lists: the same bits as gm_page_rank_multi, which
    slices the whole graph, and the same as the single-GPU engine."""
    from graph_amd import synth
    from graph_amd.distributed import partition_local_slices

    monkeypatch.setenv("GM_MULTI_ENGINE", "pb")
    scale, n = 18, 1 << 18
    src, dst = synth.rmat_edges(scale, 42)
    g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                           synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
    del src, dst
    cfg = P.PageRankConfig(12, 0.0, 0.85)
    whole, it_w, err_w = P.page_rank_multi(g, cfg, devices=[0] * ranks)
    one, _, err_1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    slices, bounds, out_full, devices = partition_local_slices(scale, 42, ranks, chunk=1 << 20)  # four chunks
    assert bounds[0] == 0 and bounds[-1] == n and sum(s.m for s in slices) == g.csr_inc.m
    got, it, err = P.page_rank_multi_slices(slices, bounds, out_full, cfg, devices)
    assert it == it_w == 12
    assert np.array_equal(got, whole) and err == err_w
    assert np.array_equal(got, one)  # exactly rounded ordinary rows + the reference's own hub sums: no partition in the bits
    # tolerance-driven stop, and bounds that are not the partitioner's (any ascending ranges are a valid partition)
    ref, it_ref, _ = P.page_rank(g, P.PageRankConfig(50, 1e-7, 0.85), P.PageRankMode.JacobiPB)
    got2, it2, _ = P.page_rank_multi_slices(slices, bounds, out_full, P.PageRankConfig(50, 1e-7, 0.85), devices)
    assert it2 == it_ref and np.array_equal(got2, ref)


@pytest.mark.parametrize("world", [2, 5])
def test_rank_local_rows_are_the_rows_of_the_whole_graph(P, world):
    """bench.py --gpus N builds every rank's rows WITHOUT the whole edge list or CSR (graph_amd/distributed.py:rank_local_rows):
    the same bounds as the reference's partitioner on the whole in-CSR (graph_ops.rs:431-439,479-509), the same out-degrees,
    and for every rank the same lists, entry for entry, as rows [lo, hi) of the whole Sorted in-CSR."""
    import torch

    from graph_amd import synth
    from graph_amd.distributed import greedy_degree_partition, pad_bounds, rank_local_rows

    scale, n = 16, 1 << 16
    src, dst = synth.rmat_edges(scale, 42)
    whole = synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted)
    off, tgt, _ = whole.host()
    want_bounds, _ = pad_bounds(greedy_degree_partition(off, world), world, n)
    want_out = torch.bincount(src, minlength=n).cpu().numpy()
    for rank in range(world):
        local, bounds, out_deg, _ = rank_local_rows(scale, 42, rank, world, 0, collective=False, chunk=1 << 18)  # four chunks
        assert np.array_equal(np.asarray(bounds), want_bounds)
        assert np.array_equal(out_deg.cpu().numpy(), want_out)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        loff, ltgt, _ = local.host()
        assert local.m == int(off[hi]) - int(off[lo])
        assert np.array_equal(np.diff(loff.astype(np.int64))[lo:hi], np.diff(off.astype(np.int64))[lo:hi])
        assert int(loff[lo]) == 0 and int(loff[hi]) == local.m  # every other row is empty
        assert np.array_equal(ltgt, tgt[int(off[lo]):int(off[hi])])


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_unsorted_layout_partitioned_gives_the_single_engines_bits_and_the_references_order(P, oracle, devices, monkeypatch):
    """CsrLayout::Unsorted through gm_page_rank_multi: every rank's slice keeps the arrival order of its rows' lists, notices
    that its hub rows are not ascending and sums them in CSR order (pb_hublong_kernel<true>) — the single-GPU engine's bits,
    and within 1e-5 of the oracle run on the same arrays (page_rank.rs:143-146)."""
    monkeypatch.setenv("GM_MULTI_ENGINE", "pb")
    scale, n = 17, 1 << 17
    s, d = oracle.rmat_edges(scale, seed=5)
    g = P.DirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Unsorted),
                           P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Unsorted), P.CsrLayout.Unsorted)
    ioff, itgt, _ = g.csr_inc.host()
    deg = np.diff(ioff.astype(np.int64))
    assert int((deg >= 4096).sum()) >= 10
    cfg = P.PageRankConfig(200, 1e-10, 0.85)
    one, it1, _ = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    got, it, _ = P.page_rank_multi(g, cfg, devices=devices)
    assert it == it1 and np.array_equal(got, one)
    ref, _, _ = oracle.page_rank_chunked(ioff, itgt, g.csr_out.degrees().astype(np.uint32), 200, 1e-10, 0.85)
    rel = np.abs(got.astype(np.float64) - ref) / ref
    print(f"Unsorted, {len(devices)} virtual ranks: max rel {rel.max():.2e}, hub rows {rel[deg >= 4096].max():.2e}")
    assert rel.max() <= 1e-5 and rel[deg >= 4096].max() <= 2e-6


def test_pieces_with_targets_beyond_n_are_refused(P):
    """gm_page_rank_multi_slices indexes an n-sized node map with the pieces' targets: a piece that names a node >= n
    (built over another graph, or with local ids) is GM_ERR_RANGE, not a memory fault."""
    import torch

    from graph_amd._lib import GraphMI355XError

    import ctypes as C

    from graph_amd._lib import check, lib, vp

    def rows(whole, lo, hi):
        h = vp()
        check(lib().gm_csr_slice_rows(whole.handle, lo, hi, None, 0, 0, C.byref(h)))
        return P.DeviceCsr(h)

    n = 8
    src, dst = np.array([5, 6, 7, 0, 1], np.uint32), np.array([0, 1, 3, 4, 6], np.uint32)
    whole = P.DeviceCsr.from_edges(n, src, dst, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    out_deg = torch.from_numpy(np.bincount(src, minlength=n).astype(np.int32)).cuda()
    good = [rows(whole, 0, 4), rows(whole, 4, 8)]
    got, it, _ = P.page_rank_multi_slices(good, [0, 4, 8], [out_deg, out_deg], P.PageRankConfig(3, 0.0, 0.85), [0, 0])
    assert it == 3 and got.shape == (n,)
    # rows 4..7 of ANOTHER graph (16 nodes) with an in-neighbour 12: not a piece of an 8-node graph
    other = P.DeviceCsr.from_edges(16, np.array([12, 1], np.uint32), np.array([5, 6], np.uint32), None, P.Direction.Incoming,
                                   P.CsrLayout.Sorted)
    with pytest.raises(GraphMI355XError, match="names node 12"):
        P.page_rank_multi_slices([good[0], rows(other, 4, 8)], [0, 4, 8], [out_deg, out_deg], P.PageRankConfig(3, 0.0, 0.85), [0, 0])


def test_real_collectives_on_two_gpus(P, oracle):
    """The RCCL path with more than one rank (grouped broadcasts per region on exchange streams): needs >= 2 GPUs, which the
    boxes of this pool do not have — kept so that the first multi-GPU run exercises it."""
    import graph_amd

    if graph_amd.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL exchange with > 1 rank cannot run here")
    g, _ = _graph(P, oracle, 18)
    cfg = P.PageRankConfig(9, 0.0, 0.85)
    one, _, err1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    two, it, err = P.page_rank_multi(g, cfg, devices=[0, 1])
    assert it == 9 and np.array_equal(two, one)


def test_one_process_per_rank_over_gloo_on_one_gpu_gives_the_single_engines_bits():
    """Front (b) as the driver launches it — torch.distributed.run, one process per rank, bench.py's partition-local
    construction and PiecewiseExchange with the in-order schedule — with the 8 ranks on ONE GPU and gloo standing in for RCCL
    (tools/debug_multi_gloo.py): after every sweep the summed error, after the last one every row's score, equal to the single
    engine's on the whole graph.  (The stream-per-part schedule of rounds 2-5 failed this in most runs and was removed in
    round 6: DESIGN.md §6, profiles/r06_streams_repro.txt.)  A launch that does
    not come up (port taken, no room for 8 processes) skips; only a run that reports differences fails."""
    import json, os, random, signal, subprocess, sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + random.randrange(400)), os.path.join(root, "tools", "debug_multi_gloo.py"),
           "--scale", "22", "--sweeps", "12", "--streams", "0"]
    proc = subprocess.Popen(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # the launcher and its 8 workers: the session this test started, nothing else
        proc.communicate()
        pytest.skip("8 gloo processes did not finish in 240 s")
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        pytest.skip(f"launch failed (rc {proc.returncode}): {stderr[-300:]}")
    rec = json.loads(lines[-1])
    assert rec["world"] == 8 and rec["streams"] == 0 and rec["sweeps"] == 12
    assert rec["rows_that_differ"] == 0 and rec["first_sweep_whose_error_differs"] is None, rec
