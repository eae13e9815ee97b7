"""world_size-2 (and 3) gloo runs of the multi-GPU PageRank driver on CPU: partition, padded
all-gather exchange, error all-reduce and the stop rule — with an oracle-backed stand-in for the
local HIP sweep (the oracle is the checker here, never the product)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _OracleRankEngine:
    """Per-rank sweep over rows [lo, hi) reading the padded rank-major x vector."""

    def __init__(self, O, ioff, itgt, od, bounds, stride, rank, damping, node_map=None):
        self.O, self.damping = O, damping
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        self.lo, self.hi, self.n = lo, hi, ioff.size - 1
        self.off = (ioff[lo:hi + 1] - ioff[lo]).astype(np.uint32)
        tg = itgt[ioff[lo]:ioff[hi]]
        if node_map is None:  # padded exchange of every row
            part = np.searchsorted(bounds[1:], tg, side="right")
            self.tgt = (part * stride + (tg - bounds[part])).astype(np.uint32)
        else:  # compact exchange: only nodes with out-edges own a slot
            self.tgt = node_map[tg].astype(np.uint32)
            assert (node_map[tg] >= 0).all()
        self.od = od[lo:hi].copy()

    def init(self, scores, x_loc):
        init = np.float32(1.0) / np.float32(self.n)
        scores[: self.hi - self.lo] = float(init)
        with np.errstate(divide="ignore"):
            x_loc[: self.hi - self.lo] = torch.from_numpy((init / self.od.astype(np.float32)).astype(np.float32))

    def sweep(self, x_in, x_out_local, scores, err):
        nl = self.hi - self.lo
        base = (np.float32(1.0) - np.float32(self.damping)) / np.float32(self.n)
        xin = x_in.numpy()
        sc = scores.numpy()
        e = 0.0
        out = np.zeros(nl, np.float32)
        for u in range(nl):
            ssum = np.float32(0)
            for i in range(self.off[u], self.off[u + 1]):
                ssum = np.float32(ssum + xin[self.tgt[i]])
            nw = np.float32(base + np.float32(np.float32(self.damping) * ssum))
            e += abs(float(np.float32(nw - sc[u])))
            sc[u] = nw
            with np.errstate(divide="ignore"):
                out[u] = np.float32(nw) / np.float32(self.od[u])
        x_out_local[:nl] = torch.from_numpy(out)
        err[0] = e


def _worker(rank, world, port, scale, max_iter, tol, q, compact=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from graph_amd.distributed import (compact_exchange_layout, greedy_degree_partition, pad_bounds,
                                       page_rank_partitioned)

    s, d = O.rmat_edges(scale, seed=3)
    n = 1 << scale
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
    od = O.out_degrees_from(n, s)
    bounds, stride = pad_bounds(greedy_degree_partition(ioff, world), world, n)
    node_map, send_rows = None, None
    if compact:
        nm, counts, stride, rows = compact_exchange_layout(torch.from_numpy(od.astype(np.int64)), bounds)
        node_map, send_rows = nm.numpy(), rows[rank]
        assert stride == max(counts) and int((nm >= 0).sum()) == int((od > 0).sum())
    eng = _OracleRankEngine(O, ioff, itgt, od, bounds, stride, rank, 0.85, node_map)
    scores, it, err = page_rank_partitioned(eng, n, int(bounds[rank + 1] - bounds[rank]), stride, max_iter, tol,
                                            torch.device("cpu"), send_rows=send_rows)
    q.put((rank, int(bounds[rank]), scores.numpy().copy(), it, err))
    dist.barrier()
    dist.destroy_process_group()


class _OraclePiecewiseEngine(_OracleRankEngine):
    """The same rows in pieces: sweep_bin captures the tiles it is handed (so a region consumed before its
    all-gather had landed would show up as wrong scores), sweep_accum finishes one row group."""

    def set_parts(self, row_splits):
        self.splits = [int(v) for v in row_splits]
        self.tile = None

    def sweep_bin(self, x_in, x_lo, x_hi):
        assert x_lo % self.TILE == 0 and x_hi % self.TILE == 0
        if getattr(self, "seen", None) is None or self.seen.numel() != x_in.numel():
            self.seen = torch.full_like(x_in, float("nan"))
        self.seen[x_lo:x_hi] = x_in[x_lo:x_hi]

    def sweep_accum(self, x_in, x_out_local, scores, part):
        if part == 0:
            self.err_parts = []
        lo, hi = self.splits[part], self.splits[part + 1]
        base = (np.float32(1.0) - np.float32(self.damping)) / np.float32(self.n)
        xin, sc, e = self.seen.numpy(), scores.numpy(), 0.0
        for u in range(lo, hi):
            ssum = np.float32(0)
            for i in range(self.off[u], self.off[u + 1]):
                ssum = np.float32(ssum + xin[self.tgt[i]])
            assert not np.isnan(ssum)
            nw = np.float32(base + np.float32(np.float32(self.damping) * ssum))
            e += abs(float(np.float32(nw - sc[u])))
            sc[u] = nw
            with np.errstate(divide="ignore"):
                x_out_local[u] = float(np.float32(nw) / np.float32(self.od[u]))
        self.err_parts.append(e)

    def sweep_fixup(self, x_out_local, scores, err):
        err[0] = sum(self.err_parts)
        self.seen = None  # the next sweep must propagate every region again


def _overlap_worker(rank, world, port, scale, max_iter, tol, parts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from graph_amd.distributed import (greedy_degree_partition, pad_bounds, page_rank_partitioned_overlapped,
                                       split_exchange_layout)

    s, d = O.rmat_edges(scale, seed=3)
    n = 1 << scale
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
    od = O.out_degrees_from(n, s)
    bounds, _ = pad_bounds(greedy_degree_partition(ioff, world), world, n)
    lay = split_exchange_layout(torch.from_numpy(od.astype(np.int64)), bounds, parts=parts, row_align=8, tile=16)
    nm = lay["node_map"].numpy()
    assert int((nm >= 0).sum()) == int((od > 0).sum()) and len(set(nm[nm >= 0].tolist())) == int((nm >= 0).sum())
    for k in range(parts):  # regions are whole tiles ...
        assert lay["group_off"][k] % 16 == 0 and lay["strides"][k] % 16 == 0 and lay["block"] % 16 == 0
    live = nm[nm >= 0]
    assert np.all(np.diff(live) > 0)  # ... and a node's slot ascends with its id (rank-major: what the hub rows' order needs)
    eng = _OraclePiecewiseEngine(O, ioff, itgt, od, bounds, 0, rank, 0.85, nm)
    eng.TILE = 16
    scores, it, err = page_rank_partitioned_overlapped(eng, lay, rank, int(bounds[rank + 1] - bounds[rank]), max_iter, tol,
                                                       torch.device("cpu"))
    q.put((rank, int(bounds[rank]), scores.numpy().copy(), it, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,max_iter,tol,parts", [(2, 5, 0.0, 2), (3, 20, 1e-3, 2), (2, 4, 0.0, 3)])
def test_partitioned_page_rank_overlapped_exchange_matches_single_rank(oracle, world, max_iter, tol, parts):
    scale = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, scale, max_iter, tol, parts, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1 << scale
    s, d = oracle.rmat_edges(scale, seed=3)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    init = np.float32(1.0) / np.float32(n)
    scores = np.full(n, init, np.float32)
    with np.errstate(divide="ignore"):
        outs = (init / od.astype(np.float32)).astype(np.float32)
    it = 0
    while True:
        outs, err = oracle.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, scores, outs)
        it += 1
        if err < tol or it == max_iter:
            break
    got = np.zeros(n, np.float32)
    for rank, lo, sc, it_r, err_r in results:
        got[lo:lo + sc.size] = sc
        assert it_r == it
        assert abs(err_r - err) <= 1e-9 * max(err, 1e-30) + 1e-15
    assert np.array_equal(got, scores)


def _sparse_worker(rank, world, port, scale, max_iter, tol, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from graph_amd.distributed import (greedy_degree_partition, pad_bounds, page_rank_partitioned_sparse,
                                       sparse_exchange_layout)

    s, d = O.rmat_edges(scale, seed=3)
    n = 1 << scale
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
    od = O.out_degrees_from(n, s)
    bounds, _ = pad_bounds(greedy_degree_partition(ioff, world), world, n)
    lay = sparse_exchange_layout(torch.from_numpy(ioff.astype(np.int64)), torch.from_numpy(itgt.astype(np.int64)), bounds, rank)
    nm = lay["node_map"].numpy()
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    needed = np.unique(itgt[ioff[lo]:ioff[hi]])
    assert lay["x_len"] == max(needed.size, 1) and np.array_equal(np.flatnonzero(nm >= 0), needed)
    assert sum(lay["recv_cnt"]) == needed.size and lay["x_len"] <= int((od > 0).sum())  # never more than the all-gather
    eng = _OracleRankEngine(O, ioff, itgt, od, bounds, 0, rank, 0.85, nm)
    scores, it, err = page_rank_partitioned_sparse(eng, lay, hi - lo, max_iter, tol, torch.device("cpu"))
    q.put((rank, lo, scores.numpy().copy(), it, err, lay["x_len"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,max_iter,tol", [(2, 5, 0.0), (3, 20, 1e-3), (4, 3, 0.0)])
def test_partitioned_page_rank_sparse_exchange_matches_single_rank(oracle, world, max_iter, tol):
    """The opt-in sparse exchange: every pair of ranks exchanges only the out_scores the receiver's rows read."""
    scale = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, scale, max_iter, tol, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1 << scale
    s, d = oracle.rmat_edges(scale, seed=3)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    init = np.float32(1.0) / np.float32(n)
    scores = np.full(n, init, np.float32)
    with np.errstate(divide="ignore"):
        outs = (init / od.astype(np.float32)).astype(np.float32)
    it = 0
    while True:
        outs, err = oracle.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, scores, outs)
        it += 1
        if err < tol or it == max_iter:
            break
    got = np.zeros(n, np.float32)
    for rank, lo, sc, it_r, err_r, x_len in results:
        got[lo:lo + sc.size] = sc
        assert it_r == it
        assert abs(err_r - err) <= 1e-9 * max(err, 1e-30) + 1e-15
    assert np.array_equal(got, scores)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,max_iter,tol,compact", [(2, 4, 0.0, False), (3, 20, 1e-3, False), (2, 5, 0.0, True),
                                                        (3, 20, 1e-3, True)])
def test_partitioned_page_rank_matches_single_rank_jacobi(oracle, world, max_iter, tol, compact):
    scale = 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scale, max_iter, tol, q, compact)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1 << scale
    s, d = oracle.rmat_edges(scale, seed=3)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    od = oracle.out_degrees_from(n, s)
    # single-rank synchronous reference with the same stop rule
    init = np.float32(1.0) / np.float32(n)
    scores = np.full(n, init, np.float32)
    with np.errstate(divide="ignore"):
        outs = (init / od.astype(np.float32)).astype(np.float32)
    it = 0
    while True:
        outs, err = oracle.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, scores, outs)
        it += 1
        if err < tol or it == max_iter:
            break
    got = np.zeros(n, np.float32)
    for rank, lo, sc, it_r, err_r in results:
        got[lo:lo + sc.size] = sc
        assert it_r == it
        assert abs(err_r - err) <= 1e-9 * max(err, 1e-30) + 1e-15
    assert np.array_equal(got, scores)  # same per-row order -> bit-exact across the partition


def test_pad_bounds_and_partition_edge_cases():
    from graph_amd.distributed import greedy_degree_partition, pad_bounds

    off = np.array([0, 100, 100, 100, 101], np.uint32)
    r = greedy_degree_partition(off, 4)
    assert r[0] == (0, 1) and r[-1][1] == 4 and len(r) <= 4
    b, stride = pad_bounds(r, 4, 4)
    assert b[0] == 0 and b[-1] == 4 and stride >= 1 and len(b) == 5
    assert greedy_degree_partition(np.zeros(1, np.uint32), 2) == []


def _wcc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from graph_amd.distributed import wcc_partitioned

    s, d = O.rmat_edges(9, seed=11)
    n = 1 << 9
    lo, hi = rank * n // world, (rank + 1) * n // world
    mine = (s >= lo) & (s < hi)  # this rank owns the out-edges of its rows
    es, ed = s[mine].astype(np.int64), d[mine].astype(np.int64)

    def link_rows(labels):  # numpy stand-in for gm_wcc_link_rows: min-label hooking + full compression
        lab = labels.numpy()
        changed = True
        while changed:
            changed = False
            for a, b in zip(es, ed):
                ra, rb = a, b
                while lab[ra] != ra:
                    ra = lab[ra]
                while lab[rb] != rb:
                    rb = lab[rb]
                if ra != rb:
                    lab[max(ra, rb)] = min(ra, rb)
                    changed = True
        for v in range(n):
            r = v
            while lab[r] != r:
                r = lab[r]
            lab[v] = r

    labels = torch.arange(n, dtype=torch.int32)
    rounds = wcc_partitioned(link_rows, labels)
    q.put((rank, labels.numpy().copy(), rounds))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_wcc_driver_gloo(oracle):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wcc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s, d = oracle.rmat_edges(9, seed=11)
    n = 1 << 9
    ooff, otgt = oracle.csr_build(n, s, d, oracle.OUTGOING, oracle.SORTED)
    ioff, itgt = oracle.csr_build(n, s, d, oracle.INCOMING, oracle.SORTED)
    ref = oracle.wcc(ooff, otgt, ioff, itgt)
    for rank, lab, rounds in results:
        assert np.array_equal(lab.astype(np.uint32), ref) and 1 <= rounds <= 10


def _sssp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from graph_amd.distributed import sssp_partitioned

    s, d = O.rmat_edges(9, seed=13)
    w = O.rmat_weights(s.size, seed=44)
    n = 1 << 9
    off, tgt, wv = O.csr_build(n, s, d, O.OUTGOING, O.SORTED, w)
    start = int(np.flatnonzero(np.diff(off) > 0)[0])
    lo, hi = rank * n // world, (rank + 1) * n // world
    inf_bits = np.array([np.finfo(np.float32).max], np.float32).view(np.int32)[0]

    def relax_rows(bits):  # numpy stand-in for gm_sssp_relax_rows over rows [lo, hi)
        dv = bits.numpy().view(np.float32)
        improved = False
        for u in range(lo, hi):
            if bits[u] == inf_bits:
                continue
            for i in range(off[u], off[u + 1]):
                nd = np.float32(dv[u] + wv[i])
                if nd < dv[tgt[i]]:
                    dv[tgt[i]] = nd
                    improved = True
        return improved

    bits = torch.full((n,), int(inf_bits), dtype=torch.int32)
    bits[start] = 0
    rounds = sssp_partitioned(relax_rows, bits)
    q.put((rank, bits.numpy().view(np.float32).copy(), rounds, start))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_sssp_driver_gloo(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sssp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s, d = oracle.rmat_edges(9, seed=13)
    w = oracle.rmat_weights(s.size, seed=44)
    off, tgt, wv = oracle.csr_build(1 << 9, s, d, oracle.OUTGOING, oracle.SORTED, w)
    for rank, dv, rounds, start in results:
        assert np.array_equal(dv, oracle.delta_stepping(off, tgt, wv, start, 0.25)) and rounds >= 1
