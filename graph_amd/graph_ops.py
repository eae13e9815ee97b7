"""Host-side graph operations of the reference's prelude that are plain index arithmetic (crates/builder/src/graph_ops.rs):
the greedy range partitions (`DegreePartitionOp`, `OutDegreePartitionOp`, `InDegreePartitionOp`).  They run on a CSR's offsets —
n + 1 integers — and decide which rank owns which rows; the rows themselves never leave the device.  numpy only."""
from __future__ import annotations

import numpy as np


def greedy_node_map_partition(prefix: np.ndarray, batch_size: int, max_batches: int):
    """graph_ops.rs:479-509 over node_map(v) = prefix[v + 1] - prefix[v]: walk the nodes in order and close a range as soon as
    its sum reaches `batch_size` while fewer than max_batches - 1 ranges exist; the last range ends at the last node.
    Returns [(start, end)] (the reference's Vec<Range<NI>>).  One binary search per range instead of a pass over the nodes."""
    if max_batches < 1:
        raise ValueError("max_batches must be at least 1")  # (the reference computes max_batches - 1 in usize: a panic)
    n = prefix.size - 1
    if n <= 0:
        return []
    pre = prefix.astype(np.int64)
    ranges, start = [], 0
    while start < n:
        if len(ranges) >= max_batches - 1:
            ranges.append((start, n))
            break
        # the first node u >= start with prefix[u + 1] - prefix[start] >= batch_size
        u = int(np.searchsorted(pre, pre[start] + int(batch_size), side="left")) - 1
        u = max(u, start)
        if u >= n - 1:
            ranges.append((start, n))
            break
        ranges.append((start, u + 1))
        start = u + 1
    return ranges


def degree_partition_of_offsets(offsets: np.ndarray, concurrency: int, total=None):
    """out_degree_partition / in_degree_partition (graph_ops.rs:394-402, 431-439: batch = ceil(edge_count / concurrency)) on
    the offsets of the out- / in-CSR; degree_partition (graph_ops.rs:357-365: batch = ceil(2 edge_count / concurrency), which
    is the degree sum again) on those of an undirected CSR.  `total`: the edge count if it is not offsets[-1]."""
    if concurrency < 1:
        raise ValueError("concurrency must be at least 1")
    n = offsets.size - 1
    if n <= 0:
        return []
    total = int(offsets[n]) if total is None else int(total)
    batch = -(-total // concurrency) if total else 0
    return greedy_node_map_partition(offsets, batch, concurrency)
