"""The device arena behind the library's large buffers (graph_amd/csrc/arena.hip) and the calls that release what a handle
parked (gm_csr_trim) or what the arena holds idle (gm_trim): results must not depend on whether buffers are fresh, reused
or rebuilt, and ordinary allocations made between plan builds must stay intact (a freed virtual range that had carried
mappings used to corrupt later hipMalloc allocations on ROCm 7.0: NaN scores at scale 21-24)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    from graph_amd import prelude

    return prelude


def _graph(P, scale, seed=42):
    from graph_amd import synth

    n = 1 << scale
    src, dst = synth.rmat_edges(scale, seed)
    g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                           synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
    return g


def test_trim_releases_and_the_next_call_rebuilds_the_same_bits(P):
    g = _graph(P, 22)  # 67 M edges: the plan's key buffers and the value stream are arena buffers (>= 128 MiB)
    cfg = P.PageRankConfig(12, 0.0, 0.85)
    first, it, err = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
    assert it == 12 and np.isfinite(first).all()
    held = P.arena_info()
    assert held["pieces_created"] > 0 and held["held_bytes"] >= held["idle_bytes"] and held["held_bytes"] > 0
    again, _, err2 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)   # on the parked plan and call state
    assert np.array_equal(again, first) and err2 == err
    g.csr_inc.trim()   # plan, value stream, vectors gone; the graph stays
    g.csr_out.trim()
    after_trim = P.arena_info()
    in_use, in_use_after = held["held_bytes"] - held["idle_bytes"], after_trim["held_bytes"] - after_trim["idle_bytes"]
    assert in_use > 0 and in_use_after < in_use  # the value stream went back to the pool
    P.trim_device()    # ... and the idle pieces back to the driver
    released = P.arena_info()
    assert released["idle_bytes"] == 0 and released["held_bytes"] <= after_trim["held_bytes"]
    rebuilt, _, err3 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)  # everything built again, from fresh pieces
    assert np.array_equal(rebuilt, first) and err3 == err
    g.csr_inc.trim()   # trimming twice, or a handle that has nothing parked, is fine
    g.csr_inc.trim()
    P.trim_device(0)


def test_ordinary_allocations_survive_plan_builds_and_releases(P, monkeypatch):
    """hipMalloc'd buffers (torch tensors here) filled with a pattern before, between and after private plan builds and their
    release must read back unchanged, and every build must give the same scores."""
    import torch

    monkeypatch.setenv("GM_PB_NOCACHE", "1")  # every call builds its own plan and drops it
    g = _graph(P, 21, seed=3)
    cfg = P.PageRankConfig(6, 0.0, 0.85)
    ref = None
    guards = []
    for k in range(4):
        t = torch.full((48 << 20,), float(k + 1), dtype=torch.float32, device="cuda")  # 192 MiB from hipMalloc
        guards.append((t, float(k + 1)))
        got, it, _ = P.page_rank(g, cfg, P.PageRankMode.JacobiPB)
        assert it == 6 and np.isfinite(got).all()
        ref = got if ref is None else ref
        assert np.array_equal(got, ref)
        if k == 1:
            P.trim_device()
        for buf, val in guards:
            assert float(buf.min()) == val and float(buf.max()) == val
    torch.cuda.synchronize()


def test_other_algorithms_work_across_a_trim(P):
    from graph_amd import synth

    scale, n = 18, 1 << 18
    src, dst = synth.rmat_edges(scale, 11)
    w = synth.rmat_weights(int(src.numel()), 12)
    out = synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted, w)
    g = P.DirectedCsrGraph(out, out, P.CsrLayout.Sorted)
    start = int(np.flatnonzero(out.degrees() > 0)[0])
    d1 = P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1))
    c1 = P.wcc_afforest(g, P.WccConfig()).to_vec()
    out.trim()
    P.trim_device()
    assert np.array_equal(P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1)), d1)
    assert np.array_equal(P.wcc_afforest(g, P.WccConfig()).to_vec(), c1)


def test_a_hundred_graphs_of_varied_sizes_keep_the_arena_bounded(P):
    """A long-lived process that builds, uses and drops graphs of many different sizes: buffer sizes come in classes (at most
    12.5 % above the request), so the address ranges and pieces released by one graph serve the next — the device memory the
    arena holds and the address space it has handed out must stop growing after the first few graphs instead of following
    the number of graphs (address space is never returned to the runtime: arena.hip)."""
    from graph_amd import synth

    rng = np.random.default_rng(7)
    P.trim_device(0)
    base_va = P.arena_va_info()
    used_va, held, ref = [], [], {}
    for k in range(100):
        scale = 20 if k % 10 else 22  # every tenth graph is four times larger
        n = 1 << scale
        m = int(rng.integers(10, 17) * n + rng.integers(0, n))  # a different edge count every time: 10-17 edges per node
        src, dst = synth.rmat_edges(scale, 100 + k)
        src, dst = src[:m].contiguous(), dst[:m].contiguous()
        g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                               synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
        del src, dst
        got, it, _ = P.page_rank(g, P.PageRankConfig(3, 0.0, 0.85), P.PageRankMode.JacobiPB)
        assert it == 3 and np.isfinite(got).all()
        if k in (0, 50):  # the same graph again much later: the same bits from recycled ranges and pieces
            ref[k] = (got.copy(), m)
        del g
        va = P.arena_va_info()
        used_va.append(va["reserved_bytes"] - va["unused_bytes"] - (base_va["reserved_bytes"] - base_va["unused_bytes"]))
        held.append(P.arena_info()["held_bytes"])
    # after the first twenty graphs (every size class seen) the next eighty add little: not 4x more
    print(f"address space handed out after 20 / 100 graphs: {used_va[19] / 2**30:.1f} / {used_va[99] / 2**30:.1f} GiB; "
          f"device memory held: {held[19] / 2**30:.1f} / {held[99] / 2**30:.1f} GiB")
    assert used_va[99] <= 1.5 * used_va[19] + (8 << 30)
    assert max(held[20:]) <= 1.5 * max(held[:20]) + (4 << 30)
    for k, (want, m) in ref.items():
        scale, n = (22 if k % 10 == 0 else 20), None
        n = 1 << scale
        src, dst = synth.rmat_edges(scale, 100 + k)
        src, dst = src[:m].contiguous(), dst[:m].contiguous()
        g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                               synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
        again, _, _ = P.page_rank(g, P.PageRankConfig(3, 0.0, 0.85), P.PageRankMode.JacobiPB)
        assert np.array_equal(again, want)
        del g
    P.trim_device(0)
