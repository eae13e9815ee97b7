"""The reference prelude's partition operations (crates/builder/src/graph_ops.rs) on the host side: the reference's own unit
tests of greedy_node_map_partition (graph_ops.rs:673-708), the doc examples of the three *PartitionOp traits through the graph
classes (on stand-in CSR handles: the operations only read offsets), and random offsets against the oracle and a literal
restatement of the reference's loop."""
import numpy as np
import pytest

from graph_amd.graph_ops import degree_partition_of_offsets, greedy_node_map_partition


def _prefix(values):
    return np.concatenate([[0], np.cumsum(np.asarray(values, dtype=np.int64))]).astype(np.uint64)


def _loop(values, batch_size, max_batches):
    """graph_ops.rs:479-509, statement by statement"""
    parts, size, start = [], 0, 0
    for node, v in enumerate(values):
        size += int(v)
        if (len(parts) < max_batches - 1 and size >= batch_size) or node == len(values) - 1:
            parts.append((start, node + 1))
            size, start = 0, node + 1
    return parts


@pytest.mark.parametrize("values,batch,max_batches,want", [
    ([1] * 10, 10, 99999, [(0, 10)]),                                                       # graph_ops.rs:673-678
    ([x % 2 for x in range(10)], 4, 99999, [(0, 8), (8, 10)]),                              # :680-686
    (list(range(10)), 6, 99999, [(0, 4), (4, 6), (6, 7), (7, 8), (8, 9), (9, 10)]),         # :688-698
    (list(range(10)), 6, 3, [(0, 4), (4, 6), (6, 10)]),                                     # :700-707
])
def test_the_references_unit_tests_of_the_greedy_partition(values, batch, max_batches, want):
    assert greedy_node_map_partition(_prefix(values), batch, max_batches) == want == _loop(values, batch, max_batches)


class _Csr:
    """stand-in for prelude.DeviceCsr: what the partition methods touch"""

    def __init__(self, n, src, dst):
        order = np.argsort(src, kind="stable")
        self.n, self.m = n, len(src)
        self._off = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]).astype(np.uint32)
        self._tgt = np.asarray(dst, np.uint32)[order]

    def host(self):
        return self._off, self._tgt, None


def _directed(P, n, edges):
    s, d = np.array([e[0] for e in edges]), np.array([e[1] for e in edges])
    return P.DirectedCsrGraph(_Csr(n, s, d), _Csr(n, d, s), P.CsrLayout.Unsorted)


def test_the_doc_examples_of_the_partition_traits():
    from graph_amd import prelude as P

    g = _directed(P, 4, [(0, 1), (0, 2), (2, 1), (2, 3)])            # graph_ops.rs:381-392
    assert g.out_degree_partition(2) == [range(0, 1), range(1, 4)]
    g = _directed(P, 4, [(1, 0), (1, 2), (2, 0), (3, 2)])            # graph_ops.rs:418-429
    assert g.in_degree_partition(2) == [range(0, 1), range(1, 4)]
    e = [(0, 1), (0, 2), (0, 3), (0, 3)]                             # graph_ops.rs:344-355
    s = np.array([a for a, b in e] + [b for a, b in e]); d = np.array([b for a, b in e] + [a for a, b in e])
    u = P.UndirectedCsrGraph(_Csr(4, s, d), P.CsrLayout.Unsorted)
    assert u.edge_count() == 4 and u.degree_partition(2) == [range(0, 1), range(1, 4)]
    assert len(u.degree_partition(1)) == 1 and u.degree_partition(1)[0] == range(0, 4)
    with pytest.raises(ValueError):
        u.degree_partition(0)


@pytest.mark.parametrize("seed", range(8))
def test_random_degree_sequences_against_the_oracle_and_the_loop(seed, oracle):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    shape = seed % 4
    deg = (rng.integers(0, 5, n) if shape == 0 else rng.integers(0, 2, n) * rng.integers(0, 300, n) if shape == 1
           else np.zeros(n, np.int64) if shape == 2 else (rng.pareto(1.2, n) * 3).astype(np.int64))
    off = _prefix(deg).astype(np.uint32)
    for conc in (1, 2, 3, 8, 64, n + 5):
        want = oracle.greedy_degree_partition(off, conc)
        got = degree_partition_of_offsets(off, conc)
        total = int(off[-1])
        batch = -(-total // conc) if total else 0
        assert got == want == _loop(deg.tolist(), batch, conc)
        assert len(got) <= conc and got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    for batch in (0, 1, 7, 1000):
        for mb in (1, 2, 5, 1000):
            assert greedy_node_map_partition(off, batch, mb) == _loop(deg.tolist(), batch, mb)


def test_empty_graph_has_no_ranges():
    assert degree_partition_of_offsets(np.zeros(1, np.uint32), 4) == []
    assert greedy_node_map_partition(np.zeros(1, np.uint32), 3, 2) == []


def test_node_values_set_the_node_count(monkeypatch):
    """csr.rs:1221-1261 (directed_ / undirected_from_node_values_exceeding_edge_list_max_id) at the builder: the values' number is
    the node count handed to the device build; fewer values than the edge list needs is the reference's panic (csr.rs:556-562).
    The device build is replaced by the stand-in CSR: what is checked here is the host logic in front of it."""
    from graph_amd import prelude as P

    seen = []

    def from_edges(cls, n, src, dst, weights, direction, layout, device=0):
        seen.append((n, int(direction)))
        if int(direction) == int(P.Direction.Incoming):
            return _Csr(n, dst, src)
        if int(direction) == int(P.Direction.Undirected):
            return _Csr(n, np.concatenate([src, dst]), np.concatenate([dst, src]))
        return _Csr(n, src, dst)

    monkeypatch.setattr(P.DeviceCsr, "from_edges", classmethod(from_edges))
    g = P.GraphBuilder().edges([(0, 1), (1, 2)]).node_values([0, 1, 2, 3]).build(P.DirectedCsrGraph)
    assert g.node_count() == 4 and [n for n, _ in seen] == [4, 4]
    assert [g.node_value(v) for v in range(4)] == [0, 1, 2, 3]
    assert [g.out_degree(v) for v in range(4)] == [1, 1, 0, 0]
    u = P.GraphBuilder().edges([(0, 1), (1, 2)]).node_values(["a", "b", "c", "d"]).build(P.UndirectedCsrGraph)
    assert u.node_count() == 4 and [u.degree(v) for v in range(4)] == [1, 2, 1, 0] and u.node_value(3) == "d"
    with pytest.raises(ValueError, match=r"number of node values \(2\) does not match node count of edge list \(3\)"):
        P.GraphBuilder().edges([(0, 1), (1, 2)]).node_values([0, 1]).build(P.DirectedCsrGraph)
    with pytest.raises(ValueError):
        P.GraphBuilder().node_values([1])
    plain = P.GraphBuilder().edges([(0, 1)]).build(P.DirectedCsrGraph)
    assert plain.node_count() == 2 and plain.node_value(1) is None
    with pytest.raises(IndexError):
        plain.node_value(2)
