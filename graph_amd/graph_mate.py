"""`graph_mate`-shaped Python front end (crates/mate/graph_mate.pyi:46-198) over the MI355X C ABI.

The reference's PyO3 module exposes `DiGraph` / `Graph` (32-bit node ids), `Layout`, `FileFormat` and
result objects; this module mirrors that surface so the reference's own pytest suite
(crates/mate/tests/*.py) runs nearly verbatim as an acceptance suite (tests/test_gpu_graph_mate.py).  Graph data lives
in HBM (graph_amd.prelude.DeviceCsr); neighbour accessors return numpy views of a lazily downloaded
host mirror, like the reference's zero-copy views (crates/mate/src/graphs/digraph.rs:126-160).
"""
from __future__ import annotations

import enum
import time

import numpy as np

from . import prelude as P


class Layout(enum.Enum):
    """graph_mate.pyi:7-28"""
    Sorted = "Sorted"
    Unsorted = "Unsorted"
    Deduplicated = "Deduplicated"


class FileFormat(enum.Enum):
    """graph_mate.pyi:30-44"""
    Graph500 = "Graph500"
    EdgeList = "EdgeList"


_LAYOUT = {Layout.Sorted: P.CsrLayout.Sorted, Layout.Unsorted: P.CsrLayout.Unsorted,
           Layout.Deduplicated: P.CsrLayout.Deduplicated, None: P.CsrLayout.Unsorted}
_LAYOUT_BACK = {v: k for k, v in _LAYOUT.items() if k is not None}


def _builder(layout):
    return P.GraphBuilder().csr_layout(_LAYOUT[layout])


def _from_file(path, layout, file_format, kind):
    fmt = P.Graph500Input() if file_format == FileFormat.Graph500 else P.EdgeListInput()
    return _builder(layout).file_format(fmt).path(path).build(kind)


def _from_array(arr, layout, kind):
    a = np.asarray(arr)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("expected an array of shape (edges, 2)")
    return _builder(layout).edges(a.astype(np.uint64)).build(kind)


def _timed(prev_micros, fn):
    """crates/mate/src/graphs/mod.rs:400-432 (`time` / `timed`): the call's result and its wall time in microseconds added to
    what the graph has cost so far"""
    t = time.perf_counter()
    out = fn()
    return out, int(prev_micros) + int((time.perf_counter() - t) * 1e6)


def _duration_debug(micros: int) -> str:
    """`{:?}` of core::time::Duration::from_micros(micros), the way the reference's __repr__ prints `load_took`
    (crates/mate/src/graphs/mod.rs:374-383): the largest unit that leaves an integer part, fraction without trailing zeros"""
    nanos_total = int(micros) * 1000
    secs, nanos = divmod(nanos_total, 1_000_000_000)

    def dec(integer, frac, width, unit):
        digits = f"{frac:0{width}d}".rstrip("0")
        return f"{integer}.{digits}{unit}" if digits else f"{integer}{unit}"

    if secs > 0:
        return dec(secs, nanos, 9, "s")
    if nanos >= 1_000_000:
        return dec(nanos // 1_000_000, nanos % 1_000_000, 6, "ms")
    if nanos >= 1_000:
        return dec(nanos // 1_000, nanos % 1_000, 3, "\u00b5s")
    return f"{nanos}ns"


class _GraphRepr:
    """what DiGraph and Graph share: `load_micros` (#[pyo3(get)], digraph.rs:20-22 / graph.rs:17-19) and the Debug form of
    PyGraph as __repr__ (mod.rs:279-281, 374-383)"""
    _load_micros = 0

    @property
    def load_micros(self) -> int:
        return self._load_micros

    def __repr__(self):
        return (f"Graph {{ node_count: {self.node_count()}, edge_count: {self.edge_count()}, "
                f"load_took: {_duration_debug(self._load_micros)} }}")


class _Timed:
    def __init__(self, micros):
        self._micros = max(int(micros), 1)

    @property
    def micros(self) -> int:
        return self._micros


class PageRankResult(_Timed):
    """crates/mate/src/page_rank.rs:42-75"""

    def __init__(self, scores, iterations, error, micros):
        super().__init__(micros)
        self._scores, self._iterations, self._error = scores, iterations, error

    def scores(self):
        return self._scores

    @property
    def ran_iterations(self) -> int:
        return self._iterations

    @property
    def error(self) -> float:
        return self._error

    def __repr__(self):
        return f"PageRankResult(ran_iterations={self._iterations}, error={self._error}, micros={self.micros})"


class WccResult(_Timed):
    def __init__(self, components, micros):
        super().__init__(micros)
        self._components = components

    def components(self):
        return self._components

    def __repr__(self):
        return f"WccResult(micros={self.micros})"


class TriangleCountResult(_Timed):
    def __init__(self, triangles, micros):
        super().__init__(micros)
        self._triangles = triangles

    @property
    def triangles(self) -> int:
        return self._triangles

    def __repr__(self):
        return f"TriangleCountResult(triangles={self._triangles}, micros={self.micros})"


class DiGraph(_GraphRepr):
    """A directed graph using 32 bits for node ids (graph_mate.pyi:46-117)."""

    def __init__(self, inner: P.DirectedCsrGraph, load_micros: int = 0):
        self._g = inner
        self._load_micros = int(load_micros)

    @staticmethod
    def load(path, layout: Layout = Layout.Unsorted, file_format=FileFormat.Graph500) -> "DiGraph":
        return DiGraph(*_timed(0, lambda: _from_file(path, layout, file_format, P.DirectedCsrGraph)))

    @staticmethod
    def from_numpy(np_array, layout: Layout = Layout.Unsorted) -> "DiGraph":
        return DiGraph(*_timed(0, lambda: _from_array(np_array, layout, P.DirectedCsrGraph)))

    @staticmethod
    def from_pandas(df, layout: Layout = Layout.Unsorted) -> "DiGraph":
        return DiGraph(*_timed(0, lambda: _from_array(df.iloc[:, :2].to_numpy(), layout, P.DirectedCsrGraph)))

    def node_count(self) -> int:
        return self._g.node_count()

    def edge_count(self) -> int:
        return self._g.edge_count()

    def out_degree(self, node: int) -> int:
        return self._g.out_degree(node)

    def in_degree(self, node: int) -> int:
        return self._g.in_degree(node)

    def out_neighbors(self, node: int):
        return self._g.out_neighbors(node)

    def in_neighbors(self, node: int):
        return self._g.in_neighbors(node)

    def copy_out_neighbors(self, node: int):
        return self._g.out_neighbors(node).tolist()

    def copy_in_neighbors(self, node: int):
        return self._g.in_neighbors(node).tolist()

    def to_undirected(self, layout: Layout = None) -> "Graph":
        # (mod.rs:248: the new graph's load_micros = this graph's + the conversion)
        return Graph(*_timed(self._load_micros, lambda: self._g.to_undirected(None if layout is None else _LAYOUT[layout])))

    def page_rank(self, *, max_iterations: int = 20, tolerance: float = 1e-4,
                  damping_factor: float = 0.85) -> PageRankResult:
        t = time.perf_counter()
        scores, it, err = P.page_rank(self._g, P.PageRankConfig(max_iterations, tolerance, damping_factor))
        return PageRankResult(scores, it, err, (time.perf_counter() - t) * 1e6)

    def wcc(self, *, chunk_size: int = 16384, neighbor_rounds: int = 2, sampling_size: int = 1024) -> WccResult:
        t = time.perf_counter()
        comp = P.wcc_afforest(self._g, P.WccConfig(chunk_size, neighbor_rounds, sampling_size)).to_vec()
        return WccResult(comp, (time.perf_counter() - t) * 1e6)


class Graph(_GraphRepr):
    """An undirected graph using 32 bits for node ids (graph_mate.pyi:119-171)."""

    def __init__(self, inner: P.UndirectedCsrGraph, load_micros: int = 0):
        self._g = inner
        self._load_micros = int(load_micros)

    @staticmethod
    def load(path, layout: Layout = Layout.Unsorted, file_format=FileFormat.Graph500) -> "Graph":
        return Graph(*_timed(0, lambda: _from_file(path, layout, file_format, P.UndirectedCsrGraph)))

    @staticmethod
    def from_numpy(np_array, layout: Layout = Layout.Unsorted) -> "Graph":
        return Graph(*_timed(0, lambda: _from_array(np_array, layout, P.UndirectedCsrGraph)))

    @staticmethod
    def from_pandas(df, layout: Layout = Layout.Unsorted) -> "Graph":
        return Graph(*_timed(0, lambda: _from_array(df.iloc[:, :2].to_numpy(), layout, P.UndirectedCsrGraph)))

    def node_count(self) -> int:
        return self._g.node_count()

    def edge_count(self) -> int:
        return self._g.edge_count()

    def degree(self, node: int) -> int:
        return self._g.degree(node)

    def neighbors(self, node: int):
        return self._g.neighbors(node)

    def copy_neighbors(self, node: int):
        return self._g.neighbors(node).tolist()

    def make_degree_ordered(self):
        # (mod.rs:275: the relabelling is added to load_micros)
        _, self._load_micros = _timed(self._load_micros, self._g.make_degree_ordered)

    def global_triangle_count(self) -> TriangleCountResult:
        t = time.perf_counter()
        tri = P.global_triangle_count(self._g)
        return TriangleCountResult(tri, (time.perf_counter() - t) * 1e6)


__all__ = ["Layout", "FileFormat", "DiGraph", "Graph", "PageRankResult", "WccResult", "TriangleCountResult"]
