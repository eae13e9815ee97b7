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


def greedy_degree_partition(offsets: np.ndarray, concurrency: int):
    """in_degree_partition + greedy_node_map_partition (graph_ops.rs:431-439, 479-509): walk nodes
    in order, close a range once its degree sum reaches ceil(edge_count / concurrency) while fewer
    than concurrency-1 ranges exist; the last range ends at node_count.  Returns [(start, end)]."""
    n = offsets.size - 1
    if n == 0:
        return []
    total = int(offsets[n])
    batch = -(-total // concurrency) if total else 0
    ranges, start = [], 0
    off64 = offsets.astype(np.int64)
    while start < n:
        if len(ranges) < concurrency - 1:
            # first node u >= start with offsets[u+1] - offsets[start] >= batch
            u = int(np.searchsorted(off64, off64[start] + batch, side="left")) - 1
            u = max(u, start)
            if u >= n - 1:
                ranges.append((start, n))
                break
            ranges.append((start, u + 1))
            start = u + 1
        else:
            ranges.append((start, n))
            break
    return ranges


def pad_bounds(ranges, world_size: int, n: int):
    """ranges -> (bounds[world_size+1], stride); ranks beyond len(ranges) own an empty range."""
    bounds = [0]
    for (_, e) in ranges:
        bounds.append(e)
    while len(bounds) < world_size + 1:
        bounds.append(n)
    stride = max(1, max(bounds[i + 1] - bounds[i] for i in range(world_size)))
    return np.asarray(bounds, np.uint32), stride


def page_rank_partitioned(engine, n_global: int, n_local: int, stride: int, max_iterations: int, tolerance: float,
                          device, group=None, init_fn=None):
    """Runs the sweeps of page_rank (page_rank.rs:88-110) across the ranks of `group`.

    engine.sweep(x_in_padded, x_out_local, scores_local, err) computes this rank's rows;
    engine.init(scores_local, x_local) fills the initial values (page_rank.rs:70-81).
    Returns (scores_local, iterations, error)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    x_pad = [torch.zeros(world * stride, dtype=torch.float32, device=device) for _ in range(2)]
    x_loc = torch.zeros(stride, dtype=torch.float32, device=device)
    scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=device)
    err = torch.zeros(1, dtype=torch.float64, device=device)
    engine.init(scores, x_loc)
    dist.all_gather_into_tensor(x_pad[0], x_loc, group=group)
    iteration, error, cur = 0, 0.0, 0
    can_stop_early = tolerance > 0.0
    if max_iterations == 0 and not can_stop_early:
        raise ValueError("max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)")
    while True:
        engine.sweep(x_pad[cur], x_loc, scores, err)
        dist.all_gather_into_tensor(x_pad[1 - cur], x_loc, group=group)
        cur = 1 - cur
        iteration += 1
        last = iteration == max_iterations
        if can_stop_early or last:
            dist.all_reduce(err, op=dist.ReduceOp.SUM, group=group)
            error = float(err.item())
            if error < tolerance or last:
                break
    return scores[:n_local], iteration, error
