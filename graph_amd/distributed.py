"""1-D vertex-range partitioned PageRank: one process per GPU, torch.distributed (RCCL over xGMI).

Scheme: rank r owns the in-CSR rows of a contiguous node range chosen by the
reference's own greedy in-degree partitioner (crates/builder/src/graph_ops.rs:431-439,479-509);
the out_scores vector is replicated.  Every sweep: local pull kernel over the own rows ->
all-gather of the ranks' new out_scores slices -> (when a tolerance is set) all-reduce of the f64
error.  The gather buffer is rank-major with a fixed slot of `stride` floats per rank, and each
rank's targets are rewritten once into that padded index space (gm_csr_slice_rows), so the
all-gather output is consumed directly as the next sweep's x_in — no unpack pass.

The local sweep is injected (`engine.sweep`), so the partition / exchange / stop logic runs
unchanged under gloo on CPU in tests/test_distributed_cpu.py with an oracle-backed stand-in.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .graph_ops import degree_partition_of_offsets


def greedy_degree_partition(offsets: np.ndarray, concurrency: int):
    """in_degree_partition + greedy_node_map_partition (graph_ops.rs:431-439, 479-509): walk nodes
    in order, close a range once its degree sum reaches ceil(edge_count / concurrency) while fewer
    than concurrency-1 ranges exist; the last range ends at node_count.  Returns [(start, end)].
    (graph_amd/graph_ops.py holds the walk; the graph classes expose it under the reference's names.)"""
    return degree_partition_of_offsets(offsets, concurrency)


def pad_bounds(ranges, world_size: int, n: int):
    """ranges -> (bounds[world_size+1], stride); ranks beyond len(ranges) own an empty range."""
    bounds = [0]
    for (_, e) in ranges:
        bounds.append(e)
    while len(bounds) < world_size + 1:
        bounds.append(n)
    stride = max(1, max(bounds[i + 1] - bounds[i] for i in range(world_size)))
    return np.asarray(bounds, np.uint32), stride


def compact_exchange_layout(out_degree, bounds):
    """Only nodes WITH out-edges are ever gathered (out_scores of the others is +inf and unused,
    page_rank.rs:78,158), so only those are exchanged: node v of rank p becomes slot
    p*stride + (number of out-degree>0 nodes of rank p before v).  out_degree: 1-D integer torch tensor
    (any device), bounds: world+1 node ids.  Returns (map int32[n] with -1 for nodes that are never a
    source, counts per rank, stride, and per rank the local row indices to send)."""
    import torch

    has_out = out_degree > 0
    world = len(bounds) - 1
    n = out_degree.numel()
    node_map = torch.full((n,), -1, dtype=torch.int32, device=out_degree.device)
    counts, send_rows = [], []
    for p in range(world):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        rows = torch.nonzero(has_out[lo:hi], as_tuple=False).flatten()
        counts.append(int(rows.numel()))
        send_rows.append(rows)
    stride = max(1, max(counts))
    for p in range(world):
        lo = int(bounds[p])
        node_map[lo + send_rows[p]] = (p * stride + torch.arange(counts[p], device=out_degree.device)).to(torch.int32)
    return node_map, counts, stride, send_rows


def sparse_exchange_layout(in_offsets, in_targets, bounds, rank: int, edges=None):
    """Layout of the *sparse* exchange for rank `rank` (opt-in): a rank only needs the
    out_scores of the sources that actually occur in its rows' in-lists — on RMAT about 0.82 / 0.64 / 0.49 of what
    the all-gather delivers at 2 / 4 / 8 ranks — so every pair of ranks exchanges exactly that list.
    in_offsets / in_targets: the whole in-CSR as 1-D integer torch tensors (any device), or edges=(src, dst) edge
    tensors and in_offsets = in_targets = None with n = bounds[-1]; bounds: world+1 node ids.
    Returns a dict for this rank: node_map int32[n] (slot in this rank's x vector, -1 = not needed here), x_len,
    recv_off[q] / recv_cnt[q] (region of x filled by rank q; q == rank: filled locally), send_rows[q] (local rows
    whose values rank q needs, in q's slot order), own_rows (local rows this rank needs itself)."""
    world = len(bounds) - 1
    n = int(bounds[-1]) if edges is not None else in_offsets.numel() - 1
    dev = edges[0].device if edges is not None else in_targets.device
    b = torch.as_tensor(np.asarray(bounds, np.int64), device=dev)
    needs = []  # needs[r]: sorted ids of the sources rank r reads
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        if edges is not None:
            src, dst = edges
            needs.append(torch.unique(src[(dst >= lo) & (dst < hi)].to(torch.int64)))
        else:
            e0, e1 = int(in_offsets[lo]), int(in_offsets[hi])
            needs.append(torch.unique(in_targets[e0:e1].to(torch.int64)))
    mine = needs[rank]
    owner = torch.bucketize(mine, b[1:], right=True)  # rank that owns each needed source
    recv_cnt = torch.bincount(owner, minlength=world).tolist()
    recv_off = [0] * world
    for q in range(1, world):
        recv_off[q] = recv_off[q - 1] + recv_cnt[q - 1]
    node_map = torch.full((n,), -1, dtype=torch.int32, device=dev)
    node_map[mine] = torch.arange(mine.numel(), dtype=torch.int32, device=dev)  # sorted ids: regions are owner-major
    lo_r, hi_r = int(bounds[rank]), int(bounds[rank + 1])
    send_rows = []
    for q in range(world):
        nq = needs[q]
        send_rows.append((nq[(nq >= lo_r) & (nq < hi_r)] - lo_r))
    return {"node_map": node_map, "x_len": max(int(mine.numel()), 1), "recv_off": recv_off, "recv_cnt": recv_cnt,
            "send_rows": send_rows, "own_rows": send_rows[rank], "world": world, "rank": rank}


class SparseExchange:
    """Blocking pairwise exchange of exactly the out_scores each rank needs (sparse_exchange_layout): one
    compaction gather, world-1 sends and world-1 receives per sweep (batch_isend_irecv: grouped send/recv on
    RCCL, plain isend/irecv on gloo), own values copied locally."""

    def __init__(self, layout, n_local: int, device, group=None):
        self.lay, self.group, self.rank, self.world = layout, group, layout["rank"], layout["world"]
        self.x = [torch.zeros(layout["x_len"], dtype=torch.float32, device=device) for _ in range(2)]
        self.x_loc = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
        self.rows = [r.to(device) for r in layout["send_rows"]]
        self.peers = [q for q in range(self.world) if q != self.rank]
        self.send = {q: torch.zeros(max(int(self.rows[q].numel()), 1), dtype=torch.float32, device=device) for q in self.peers}
        self.cur = 0

    def exchange(self, buf: int):
        lay, x = self.lay, self.x[buf]
        o, c = lay["recv_off"][self.rank], lay["recv_cnt"][self.rank]
        x[o:o + c] = self.x_loc[self.rows[self.rank]]
        ops = []
        for q in self.peers:
            k = int(self.rows[q].numel())
            if k:
                self.send[q][:k] = self.x_loc[self.rows[q]]
                ops.append(dist.P2POp(dist.isend, self.send[q][:k], q, self.group))
            o, c = lay["recv_off"][q], lay["recv_cnt"][q]
            if c:
                ops.append(dist.P2POp(dist.irecv, x[o:o + c], q, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


def page_rank_partitioned_sparse(engine, layout, n_local: int, max_iterations: int, tolerance: float, device,
                                 group=None):
    """page_rank (page_rank.rs:88-110) across the ranks of `group` with the sparse exchange; engine.sweep(x_in,
    x_out_local, scores_local, err) reads x through layout["node_map"].  Same results as the other drivers."""
    if max_iterations == 0 and not tolerance > 0.0:
        raise ValueError("max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)")
    scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
    err = torch.zeros(1, dtype=torch.float64, device=device)
    ex = SparseExchange(layout, n_local, device, group)
    engine.init(scores, ex.x_loc)
    ex.exchange(0)
    iteration, error, cur = 0, 0.0, 0
    can_stop_early = tolerance > 0.0
    while True:
        engine.sweep(ex.x[cur], ex.x_loc, scores, err)
        ex.exchange(1 - cur)
        cur = 1 - cur
        iteration += 1
        last = iteration == max_iterations
        if can_stop_early or last:
            dist.all_reduce(err, op=dist.ReduceOp.SUM, group=group)
            error = float(err.item())
            if error < tolerance or last:
                break
    return scores[:n_local], iteration, error


ROW_ALIGN = 16384  # row splits of a sweep in pieces: a multiple of the rows per bin of any plan (<= 16384)
SOURCE_TILE = 32768  # x regions of a sweep in pieces: a multiple of any plan's source tile (gm_pr_part_geometry)


def split_exchange_layout(out_degree, bounds, parts: int = 2, row_align: int = ROW_ALIGN, tile: int = SOURCE_TILE):
    """compact_exchange_layout for a sweep in pieces: every rank cuts its rows into `parts` groups (at multiples of
    row_align) and the exchanged vector into as many regions, region k holding group k of every rank, so that region k can
    be all-gathered while the ranks still work on group k+1 and be consumed while region k+1 is still in flight.

    The vector is RANK-MAJOR (round 5; the C ABI front, multi.hip, has been from the start): rank p's block of `block` =
    sum(strides) floats holds its groups one behind the other (group k: `strides[k]` floats, a multiple of `tile`), so a
    node's slot ascends with its id and a Sorted row of a rank's slice stays ascending in the exchange index space — the
    order the hub rows' left-to-right sums follow (page_rank.rs:143-146).  Region k is therefore not one stretch of x but P
    of them: elements [p * block + group_off[k], + strides[k]) for every rank p (`region_ranges(layout, k)`).  (Until round 5
    the vector was REGION-major — what one all_gather_into_tensor per region writes — and a slice's hub rows were summed in
    an order that was neither the reference's nor the single-GPU run's.)

    Returns a dict: node_map int32[n] (-1: never a source), x_len, strides[k], group_off[k], block, world,
    row_splits[rank] (parts+1 local row indices), send_rows[rank][k] (local rows, slot order)."""
    has_out = out_degree > 0
    world = len(bounds) - 1
    n = out_degree.numel()
    dev = out_degree.device
    node_map = torch.full((n,), -1, dtype=torch.int32, device=dev)
    row_splits, send_rows = [], []
    for p in range(world):
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        n_loc = hi - lo
        sp = [min(n_loc, -(-(n_loc * k // parts) // row_align) * row_align) for k in range(parts)] + [n_loc]
        row_splits.append(sp)
        send_rows.append([sp[k] + torch.nonzero(has_out[lo + sp[k]:lo + sp[k + 1]], as_tuple=False).flatten()
                          for k in range(parts)])
    strides, group_off, off = [], [], 0
    for k in range(parts):
        most = max(int(send_rows[p][k].numel()) for p in range(world))
        stride = max(tile, -(-most // tile) * tile)
        strides.append(stride)
        group_off.append(off)
        off += stride
    block = off
    for p in range(world):
        lo = int(bounds[p])
        for k in range(parts):
            rows = send_rows[p][k]
            node_map[lo + rows] = (p * block + group_off[k] + torch.arange(rows.numel(), device=dev)).to(torch.int32)
    return {"node_map": node_map, "x_len": world * block, "strides": strides, "group_off": group_off, "block": block,
            "world": world, "row_splits": row_splits, "send_rows": send_rows, "parts": parts}


def source_flags(node_map: torch.Tensor, no_in_edges: torch.Tensor, x_len: int):
    """uint8[x_len] for DeviceCsr.set_source_flags (gm_csr_set_source_flags): 1 where the node in that slot of the exchanged vector
    has at most ONE in-edge.  node_map int32[n] (-1: never a source), no_in_edges bool / uint8 [n]: in-degree <= 1, over the GLOBAL
    in-degrees (a rank's slice cannot see them in its own offsets).  With the flags a slice's propagation-blocking plan flags the rows the whole graph's
    plan flags — rows that sum many constant terms are summed the reference's way (GM_PB_HUB_LEAVES)."""
    flags = torch.zeros(max(int(x_len), 1), dtype=torch.uint8, device=node_map.device)
    sel = (node_map >= 0) & no_in_edges.to(node_map.device).bool()
    flags[node_map[sel].long()] = 1
    return flags


def region_ranges(layout, k: int):
    """[(lo, hi)] of region k of a split_exchange_layout vector: one stretch per rank"""
    return [(p * layout["block"] + layout["group_off"][k], p * layout["block"] + layout["group_off"][k] + layout["strides"][k])
            for p in range(layout["world"])]


class PiecewiseExchange:
    """The per-sweep schedule of a partitioned PageRank whose exchange overlaps the work:

        wait region 0 -> propagate its tiles -> wait region 1 -> propagate its tiles -> ...
        rows of group 0 -> start all-gather of region 0 -> rows of group 1 -> start all-gather of region 1 ...

    so the all-gather of region k runs (RCCL, its own stream) under the accumulate of group k+1 and under
    the propagation of regions < k of the next sweep.  engine: sweep_bin / sweep_accum / sweep_fixup /
    set_parts (graph_amd.engine.PageRankEngine, or a stand-in with the same methods)."""

    def __init__(self, engine, layout, rank: int, n_local: int, device, group=None, gather=None, split_bin=True, x=None):
        # split_bin=False: one propagation launch after every region has landed (only the accumulate is cut into
        # row groups) — region k still travels under the accumulate of the later groups, and the short kernels
        # of a many-rank run are not cut in four
        self.split_bin = split_bin
        # Every part runs on the caller's stream, in order.  Until round 5 there was a second schedule with part k on a
        # HIP stream of its own (10-15 % per emulated rank at 4 and 8 ranks); with one PROCESS per rank sharing a GPU over
        # gloo it computed wrong intermediate sweeps in two runs of three (profiles/r05_multi_rank_streams.txt) — a consumer
        # overtaking its producer across streams although an event ordered them.  Round 6: the same event graph as a
        # torch-free HIP program with stamping kernels, 8-16 processes on one GPU, held in 30 runs of 30 x 300 sweeps
        # (tools/streams_repro.hip, profiles/r06_streams_repro.txt) while this class still failed beside it on the same box:
        # the fault is not in the HIP event graph itself and was not found; the schedule was REMOVED, not fixed.
        self.engine, self.layout, self.rank, self.group = engine, layout, rank, group
        world = len(layout["row_splits"])
        self.parts = layout["parts"]
        # x: the two exchanged vectors (tests with virtual ranks on one device hand every rank the SAME pair)
        self.x = x if x is not None else [torch.zeros(layout["x_len"], dtype=torch.float32, device=device) for _ in range(2)]
        self.x_loc = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
        self.x_send = [torch.zeros(s, dtype=torch.float32, device=device) for s in layout["strides"]]
        self.send_rows = [r.to(device) for r in layout["send_rows"][rank]]
        # region k = one stretch per rank (rank-major vector): the all-gather of a region lands in P views of x
        self.ranges = [region_ranges(layout, k) for k in range(self.parts)]
        self.views = [[[x[lo:hi] for (lo, hi) in self.ranges[k]] for k in range(self.parts)] for x in self.x]
        self._region_launch = hasattr(engine, "set_bin_regions")  # one propagation launch per region (graph_amd.engine)
        if self._region_launch:
            flat = [(lo, hi, k) for k in range(self.parts) for (lo, hi) in self.ranges[k]]
            engine.set_bin_regions([f[0] for f in flat], [f[1] for f in flat], [f[2] for f in flat], self.parts)
        self.cur = 0
        self.works = [None] * self.parts
        # gather(dst_views, src, k): stand-in for the collective (single-process emulation: dst_views[p] is where rank p's
        # group k lands); default RCCL / gloo
        self._gather = gather
        engine.set_parts(layout["row_splits"][rank])

    def _start_gather(self, buf: int, k: int):
        rows = self.send_rows[k]
        self.x_send[k][: rows.numel()] = self.x_loc[rows]  # compaction: only nodes with out-edges travel
        dst = self.views[buf][k]  # rank p's group k lands in rank p's block of x
        if self._gather is not None:
            self._gather(dst, self.x_send[k], k)
            self.works[k] = None
        else:
            self.works[k] = dist.all_gather(dst, self.x_send[k], group=self.group, async_op=True)

    def _bin_region(self, x_in, k: int):
        if self._region_launch:
            self.engine.sweep_bin_region(x_in, k)
        else:  # (stand-in engines of the gloo tests: one call per stretch)
            for lo, hi in self.ranges[k]:
                self.engine.sweep_bin(x_in, lo, hi)

    def start(self, scores: torch.Tensor):
        """page_rank.rs:70-81 initial values, then the first exchange"""
        self.engine.init(scores, self.x_loc)
        for k in range(self.parts):
            self._start_gather(self.cur, k)

    def sweep(self, scores: torch.Tensor, err: torch.Tensor, events=None):
        """events: optional list receiving a (start, end) torch.cuda.Event pair around every kernel piece
        (bench.py: kernel time without the waits for the collectives)"""
        e, x_in = self.engine, self.x[self.cur]

        def timed(fn, *a):
            if events is None:
                return fn(*a)
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            fn(*a)
            pair[1].record()
            events.append(pair)

        for k in range(self.parts):
            if self.works[k] is not None:
                self.works[k].wait()  # orders the current stream behind the collective; the host does not block
            if self.split_bin:
                timed(self._bin_region, x_in, k)
        if not self.split_bin:
            timed(e.sweep_bin, x_in, 0, x_in.numel())
        for k in range(self.parts):
            timed(e.sweep_accum, x_in, self.x_loc, scores, k)
            self._start_gather(1 - self.cur, k)
        e.sweep_fixup(self.x_loc, scores, err)
        self.cur = 1 - self.cur

    def finish(self):
        for k in range(self.parts):
            if self.works[k] is not None:
                self.works[k].wait()
                self.works[k] = None


def page_rank_partitioned_overlapped(engine, layout, rank: int, n_local: int, max_iterations: int, tolerance: float,
                                     device, group=None):
    """page_rank (page_rank.rs:88-110) across the ranks of `group` with the exchange overlapped
    (PiecewiseExchange).  Same results as page_rank_partitioned, bit for bit.
    Returns (scores_local, iterations, error)."""
    if max_iterations == 0 and not tolerance > 0.0:
        raise ValueError("max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)")
    scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
    err = torch.zeros(1, dtype=torch.float64, device=device)
    ex = PiecewiseExchange(engine, layout, rank, n_local, device, group)
    ex.start(scores)
    iteration, error = 0, 0.0
    can_stop_early = tolerance > 0.0
    while True:
        ex.sweep(scores, err)
        iteration += 1
        last = iteration == max_iterations
        if can_stop_early or last:
            dist.all_reduce(err, op=dist.ReduceOp.SUM, group=group)
            error = float(err.item())
            if error < tolerance or last:
                break
    ex.finish()
    return scores[:n_local], iteration, error


def page_rank_partitioned(engine, n_global: int, n_local: int, stride: int, max_iterations: int, tolerance: float,
                          device, group=None, send_rows=None):
    """Runs the sweeps of page_rank (page_rank.rs:88-110) across the ranks of `group`.

    engine.sweep(x_in_padded, x_out_local, scores_local, err) computes this rank's rows;
    engine.init(scores_local, x_local) fills the initial values (page_rank.rs:70-81).
    send_rows (optional): local row indices whose out_scores are exchanged, in slot order (see
    compact_exchange_layout; `stride` is then the compact stride); default: every local row.
    Returns (scores_local, iterations, error)."""
    world = dist.get_world_size(group)
    x_pad = [torch.zeros(world * stride, dtype=torch.float32, device=device) for _ in range(2)]
    compact = send_rows is not None
    x_loc = torch.zeros(max(n_local, 1) if compact else stride, dtype=torch.float32, device=device)
    x_send = torch.zeros(stride, dtype=torch.float32, device=device) if compact else x_loc
    scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
    err = torch.zeros(1, dtype=torch.float64, device=device)

    def exchange(dst):
        if compact:
            x_send[: send_rows.numel()] = x_loc[send_rows]
        dist.all_gather_into_tensor(dst, x_send, group=group)

    engine.init(scores, x_loc)
    exchange(x_pad[0])
    iteration, error, cur = 0, 0.0, 0
    can_stop_early = tolerance > 0.0
    if max_iterations == 0 and not can_stop_early:
        raise ValueError("max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)")
    while True:
        engine.sweep(x_pad[cur], x_loc, scores, err)
        exchange(x_pad[1 - cur])
        cur = 1 - cur
        iteration += 1
        last = iteration == max_iterations
        if can_stop_early or last:
            dist.all_reduce(err, op=dist.ReduceOp.SUM, group=group)
            error = float(err.item())
            if error < tolerance or last:
                break
    return scores[:n_local], iteration, error


def wcc_partitioned(link_rows, labels: torch.Tensor, group=None, max_rounds: int = 64):
    """Partitioned WCC: `labels` (int32/uint32 view of u32[n], replicated, initialised to
    0..n-1) is updated in place.  link_rows(labels) links the edges of this rank's rows into the local
    replica and compresses it; then the replicas are min-all-reduced; repeat until no rank changed
    anything.  Labels only ever decrease and every value is a node of the same component, so the loop
    ends with labels[u] = minimum node id of u's component on every rank.  Returns the number of rounds.
    A capacity path (the replicated n-vector is min-reduced every round), not a speed path: one GPU finishes
    RMAT scale-22 WCC in 0.6 ms of kernels."""
    # the collectives compare the labels as SIGNED 32-bit integers: ids >= 2^31 would order as negative and
    # break the parent[x] <= x invariant the link / compress kernels rely on
    if labels.numel() > (1 << 31):
        raise ValueError("wcc_partitioned: more than 2^31 nodes need an unsigned min-reduction (int32 view)")
    changed = torch.zeros(1, dtype=torch.int32, device=labels.device)
    for rounds in range(1, max_rounds + 1):
        before = labels.clone()
        link_rows(labels)
        dist.all_reduce(labels, op=dist.ReduceOp.MIN, group=group)
        changed[0] = int(not torch.equal(before, labels))
        dist.all_reduce(changed, op=dist.ReduceOp.MAX, group=group)
        if int(changed.item()) == 0:
            return rounds
    raise RuntimeError("wcc_partitioned did not converge")


def sssp_partitioned(relax_rows, dist_bits: torch.Tensor, group=None, max_rounds: int = 100000):
    """Partitioned SSSP: `dist_bits` (int32 view of the u32 bit patterns of non-negative
    f32 distances, replicated; f32::MAX everywhere except 0 at the start node) is updated in place.
    relax_rows(dist_bits) -> bool runs one relaxation pass over this rank's rows and tells whether any
    distance improved.  Each round: local passes to a local fixed point, then an integer min-all-reduce
    (bit patterns of non-negative floats order like the floats); stop when no rank improved anything.
    The result is the least fixed point — identical to delta_stepping.  Returns the number of rounds."""
    flag = torch.zeros(1, dtype=torch.int32, device=dist_bits.device)
    for rounds in range(1, max_rounds + 1):
        improved = False
        while relax_rows(dist_bits):
            improved = True
        before = dist_bits.clone()
        dist.all_reduce(dist_bits, op=dist.ReduceOp.MIN, group=group)
        flag[0] = int(improved or not torch.equal(before, dist_bits))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            return rounds
    raise RuntimeError("sssp_partitioned did not converge")


# ------------------------------------------------------------------------------------------------
# Partition-local construction: no device ever holds the whole graph (gm_page_rank_multi_slices)
# ------------------------------------------------------------------------------------------------
def rmat_degrees(scale: int, seed: int = 42, edge_factor: int = 16, device: int = 0, chunk: int = 1 << 26):
    """(in_degree, out_degree) of every node of the R-MAT graph, int64 on `device`, from the counter-based generator in
    chunks of `chunk` edges: n-sized vectors only, the edge list never exists as a whole.  (A job whose ranks each scan
    m / P edge indices sums these histograms with an all-reduce; here every caller scans all of them.)"""
    from . import synth

    n, m = 1 << scale, edge_factor << scale
    dev = torch.device("cuda", device)
    ind = torch.zeros(n, dtype=torch.int64, device=dev)
    outd = torch.zeros(n, dtype=torch.int64, device=dev)
    for first in range(0, m, chunk):
        src, dst = synth.rmat_edge_range(scale, seed, first, min(chunk, m - first), device)
        ind += torch.bincount(dst, minlength=n)
        outd += torch.bincount(src, minlength=n)
        del src, dst
    return ind, outd


def partition_local_slices(scale: int, seed: int, ranks: int, devices=None, edge_factor: int = 16, chunk: int = 1 << 26):
    """The pieces gm_page_rank_multi_slices takes, built WITHOUT the whole graph on any device: the reference's greedy
    in-degree ranges (graph_ops.rs:431-439,479-509) from the degree histograms, then per rank: the edges whose destination lies
    in its range (kept from a chunked scan of the generator), a Sorted in-CSR over them and its row slice.
    Returns (slices, bounds, out_degree_full_u32 per rank, devices)."""
    import ctypes as C

    from . import synth
    from ._lib import check, lib, vp
    from .prelude import CsrLayout, DeviceCsr, Direction

    devices = list(devices) if devices is not None else [0] * ranks
    n, m = 1 << scale, edge_factor << scale
    ind, outd = rmat_degrees(scale, seed, edge_factor, devices[0], chunk)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(ind.cpu().numpy(), out=off[1:])
    bounds, _ = pad_bounds(greedy_degree_partition(off, ranks), ranks, n)
    slices, out_full = [], []
    for p in range(ranks):
        dev = devices[p]
        lo, hi = int(bounds[p]), int(bounds[p + 1])
        keep_s, keep_d = [], []
        for first in range(0, m, chunk):
            src, dst = synth.rmat_edge_range(scale, seed, first, min(chunk, m - first), dev)
            mine = (dst >= lo) & (dst < hi)
            keep_s.append(src[mine])
            keep_d.append(dst[mine])
            del src, dst, mine
        s = torch.cat(keep_s) if keep_s else torch.empty(0, dtype=torch.int32, device=dev)
        d = torch.cat(keep_d) if keep_d else torch.empty(0, dtype=torch.int32, device=dev)
        del keep_s, keep_d
        # a CSR over the rank's own edges (n rows, nearly all of them empty: n + 1 offsets), then its rows [lo, hi)
        local = synth.build_csr(n, s, d, Direction.Incoming, CsrLayout.Sorted, None, dev)
        del s, d
        h = vp()
        check(lib().gm_csr_slice_rows(local.handle, lo, hi, None, 0, 0, C.byref(h)))
        slices.append(DeviceCsr(h))
        del local
        out_full.append(outd.to(torch.device("cuda", dev)).to(torch.int32).contiguous())
    return slices, [int(b) for b in bounds], out_full, devices


def rank_local_rows(scale: int, seed: int, rank: int, world: int, device: int = 0, edge_factor: int = 16, group=None,
                    collective: bool = True, chunk: int = 1 << 26):
    """One rank's share of the R-MAT graph for the one-process-per-GPU front (bench.py --gpus N), built WITHOUT the whole
    edge list or the whole CSR on any device:

      1. degree histograms: this rank scans edge indices [rank m / world, (rank + 1) m / world) of the counter-based generator
         in chunks and the ranks SUM their in- / out-degree vectors (`collective`; without a process group — the one-device
         emulation of a rank — the caller scans all of them);
      2. the reference's greedy in-degree ranges over the summed in-degrees (graph_ops.rs:431-439,479-509) — every rank computes
         the same bounds;
      3. the edges whose destination lies in this rank's range, kept from a chunked scan of the generator;
      4. a Sorted in-CSR over them (n rows, all but the rank's own empty).

    Returns (csr over the rank's edges — with `.no_in_edges`, uint8[n] over the GLOBAL in-degrees, for source_flags() —, bounds
    uint32[world + 1], out_degree int32[n] on the device, peak bytes of the edge buffers).  The caller slices rows [bounds[rank], bounds[rank + 1]) out of the CSR (gm_csr_slice_rows_map: targets rewritten
    into the exchange index space) and drops it."""
    from . import synth
    from .prelude import CsrLayout, Direction

    n, m = 1 << scale, edge_factor << scale
    dev = torch.device("cuda", device)
    ind = torch.zeros(n, dtype=torch.int64, device=dev)
    outd = torch.zeros(n, dtype=torch.int64, device=dev)
    shared = collective and world > 1
    e_lo, e_hi = (rank * m // world, (rank + 1) * m // world) if shared else (0, m)
    for first in range(e_lo, e_hi, chunk):
        src, dst = synth.rmat_edge_range(scale, seed, first, min(chunk, e_hi - first), device)
        ind += torch.bincount(dst, minlength=n)
        outd += torch.bincount(src, minlength=n)
        del src, dst
    if shared:
        dist.all_reduce(ind, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(outd, op=dist.ReduceOp.SUM, group=group)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(ind.cpu().numpy(), out=off[1:])
    no_in_edges = (ind <= 1).to(torch.uint8)  # at most ONE in-edge (global: the summed histogram) -> source_flags() for the rank's slice
    del ind
    bounds, _ = pad_bounds(greedy_degree_partition(off, world), world, n)
    del off
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    keep_s, keep_d, peak = [], [], 0
    for first in range(0, m, chunk):
        src, dst = synth.rmat_edge_range(scale, seed, first, min(chunk, m - first), device)
        mine = (dst >= lo) & (dst < hi)
        keep_s.append(src[mine])
        keep_d.append(dst[mine])
        del src, dst, mine
    s = torch.cat(keep_s) if keep_s else torch.empty(0, dtype=torch.int32, device=dev)
    d = torch.cat(keep_d) if keep_d else torch.empty(0, dtype=torch.int32, device=dev)
    del keep_s, keep_d
    peak = int(s.numel()) * 8 * 2 + min(chunk, m) * 9  # the kept edges twice (list + concatenation) + one chunk and its mask
    local = synth.build_csr(n, s, d, Direction.Incoming, CsrLayout.Sorted, None, device)
    del s, d
    torch.cuda.empty_cache()
    local.no_in_edges = no_in_edges
    return local, bounds, outd.to(torch.int32), peak
