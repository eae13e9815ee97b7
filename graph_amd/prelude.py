"""Host-side mirror of the reference's ``graph::prelude`` over the MI355X C ABI.

Same names, argument meaning and error behaviour as the Rust API
(crates/algos/src/prelude.rs:1-7 re-exporting crates/builder/src/prelude.rs), so the parity tests
read like the reference's own tests:

    g = GraphBuilder().csr_layout(CsrLayout.Sorted).edges([(0, 1), (1, 2)]).build(DirectedCsrGraph)
    scores, iterations, error = page_rank(g, PageRankConfig(10, 1e-4, 0.85))
    components = wcc_afforest(g, WccConfig()).to_vec()
    distances = delta_stepping(gw, DeltaSteppingConfig(0, 3.0))
    triangles = global_triangle_count(ug)

Graph objects own device-resident CSR handles (uploaded / built once); every algorithm runs in the
hand-written HIP kernels of graph_amd/csrc.  Where the reference panics (out-of-range start node,
empty sample set, ...) these functions raise.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import sys
import enum
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, lib, vp, u64, f64
from .graph_ops import degree_partition_of_offsets


class CsrLayout(enum.IntEnum):
    """crates/builder/src/graph/csr.rs:34-45 (default: Unsorted)."""
    Unsorted = 0
    Sorted = 1
    Deduplicated = 2


class Direction(enum.IntEnum):
    Outgoing = 0
    Incoming = 1
    Undirected = 2


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _result_buffer(count, dtype):
    """Host array for a per-node result.  Page-locked when torch is importable (its caching host allocator hands
    the same block out again on the next call): the device-to-host copy of 64 MB of distances then runs at link
    speed instead of through the runtime's staging buffer into freshly faulted pages."""
    try:
        import torch

        return torch.empty(count, dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True).numpy()
    except Exception:  # no torch, or no page-locked memory to be had: plain pageable memory works too
        return np.empty(count, dtype)


def _ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


def trim_device(device: int = -1):
    """gm_trim: hand the unused 64 MiB pieces of the library's device arena back to the driver (-1: current device)"""
    check(lib().gm_trim(int(device)))


def arena_info(device: int = -1):
    """gm_arena_info: what the device arena holds — bytes held, bytes of it idle, pieces created, pieces handed out so far"""
    out = (C.c_uint64 * 4)()
    check(lib().gm_arena_info(int(device), out))
    return {"held_bytes": int(out[0]), "idle_bytes": int(out[1]), "pieces_created": int(out[2]), "pieces_handed_out": int(out[3])}


def arena_va_info(device: int = -1):
    """gm_arena_va_info: the arena's virtual address space — bytes reserved, bytes in released ranges, bytes never handed out"""
    out = (C.c_uint64 * 3)()
    check(lib().gm_arena_va_info(int(device), out))
    return {"reserved_bytes": int(out[0]), "released_bytes": int(out[1]), "unused_bytes": int(out[2])}


class DeviceCsr:
    """One device-resident CSR (offsets/targets[/weights] in HBM); lazily mirrored to the host for
    the per-node accessors of the reference's graph traits."""

    def __init__(self, handle):
        # own a private copy of the pointer value: callers may reuse their ctypes out-parameter
        self._h = vp(handle.value if isinstance(handle, vp) else handle)
        self._host = None

    @classmethod
    def from_edges(cls, n, src, dst, weights, direction, layout, device=0):
        src, dst = _u32(src), _u32(dst)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        h = vp()
        check(lib().gm_csr_build_host(n, src.size, _ptr(src), _ptr(dst), _ptr(w), int(direction), int(layout),
                                      device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, offsets, targets, weights=None, device=0):
        offsets = np.ascontiguousarray(offsets)
        targets = np.ascontiguousarray(targets)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        h = vp()
        n, m = offsets.size - 1, targets.size
        if offsets.dtype == np.uint64 or targets.dtype == np.uint64:  # usize graphs
            o64, t64 = offsets.astype(np.uint64), targets.astype(np.uint64)
            check(lib().gm_csr_upload_u64(_ptr(o64), _ptr(t64), _ptr(w), n, m, device, C.byref(h)))
        else:
            o32, t32 = _u32(offsets), _u32(targets)
            check(lib().gm_csr_upload_u32(_ptr(o32), _ptr(t32), _ptr(w), n, m, device, C.byref(h)))
        return cls(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and sys is not None and not sys.is_finalizing():  # at interpreter exit the HIP runtime may already be gone
            try:
                lib().gm_csr_free(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    @property
    def n(self):
        return int(lib().gm_csr_node_count(self._h))

    @property
    def m(self):
        return int(lib().gm_csr_edge_count(self._h))

    @property
    def weighted(self):
        return lib().gm_csr_weights_ptr(self._h) != 0

    def host(self):
        if self._host is None:
            off = np.empty(self.n + 1, np.uint32)
            tgt = np.empty(self.m, np.uint32)
            w = np.empty(self.m, np.float32) if self.weighted else None
            check(lib().gm_csr_download(self._h, _ptr(off), _ptr(tgt) if self.m else None,
                                        _ptr(w) if (w is not None and self.m) else None))
            self._host = (off, tgt, w)
        return self._host

    def degrees(self):
        d = np.empty(self.n, np.uint32)
        check(lib().gm_csr_degrees(self._h, _ptr(d) if self.n else None))
        return d

    def set_source_flags(self, flags):
        """gm_csr_set_source_flags: for a ROW SLICE of a partitioned graph — one uint8 per entry of the exchange vector its lists index
        (a torch CUDA tensor on the handle's device, or None to remove them), non-zero = the node in that slot has at most one in-edge.  Lets
        the propagation-blocking plan flag the rows the whole graph's plan flags (GM_PB_HUB_LEAVES)."""
        if flags is None:
            check(lib().gm_csr_set_source_flags(self._h, 0, 0))
            return
        import torch

        assert flags.dtype == torch.uint8 and flags.is_cuda and flags.is_contiguous()
        check(lib().gm_csr_set_source_flags(self._h, int(flags.data_ptr()), int(flags.numel())))

    def trim(self):
        """gm_csr_trim: release what the handle parked for later calls (PageRank plan and call state, SSSP / WCC working
        sets, the triangle count's DAG, the multi-GPU state); the graph stays, the next call rebuilds what it needs."""
        check(lib().gm_csr_trim(self._h))


class _GraphBase:
    _node_values = None  # NodeValues<NV> (csr.rs:316-322); None = the reference's NV = ()

    def node_count(self):
        return self._n

    def node_value(self, u):
        """NodeValues::node_value (lib.rs:323-326, csr.rs:475-479, 692-696); a graph built without values has NV = (): None"""
        self._check_node(u)
        return None if self._node_values is None else self._node_values[u]

    def _check_node(self, u):
        if not 0 <= u < self._n:
            raise IndexError(f"node {u} out of range (node_count {self._n})")


class DirectedCsrGraph(_GraphBase):
    """DirectedCsrGraph = csr_out + csr_inc (crates/builder/src/graph/csr.rs:364-520)."""

    def __init__(self, csr_out: DeviceCsr, csr_inc: DeviceCsr, layout=CsrLayout.Unsorted, edges=None):
        self.csr_out, self.csr_inc = csr_out, csr_inc
        self._n = csr_out.n
        self.layout = layout
        self._edges = edges  # (src, dst, weights) kept for to_undirected

    def edge_count(self):
        return self.csr_out.m

    def out_degree(self, u):
        self._check_node(u)
        off = self.csr_out.host()[0]
        return int(off[u + 1] - off[u])

    def in_degree(self, u):
        self._check_node(u)
        off = self.csr_inc.host()[0]
        return int(off[u + 1] - off[u])

    def out_neighbors(self, u):
        self._check_node(u)
        off, tgt, _ = self.csr_out.host()
        return tgt[off[u]:off[u + 1]]

    def in_neighbors(self, u):
        self._check_node(u)
        off, tgt, _ = self.csr_inc.host()
        return tgt[off[u]:off[u + 1]]

    def out_neighbors_with_values(self, u):
        self._check_node(u)
        off, tgt, w = self.csr_out.host()
        return list(zip(tgt[off[u]:off[u + 1]].tolist(), w[off[u]:off[u + 1]].tolist()))

    def in_neighbors_with_values(self, u):
        self._check_node(u)
        off, tgt, w = self.csr_inc.host()
        return list(zip(tgt[off[u]:off[u + 1]].tolist(), w[off[u]:off[u + 1]].tolist()))

    def out_degree_partition(self, concurrency: int):
        """OutDegreePartitionOp (graph_ops.rs:29-38, 368-403): at most `concurrency` ranges of roughly equal total out-degree;
        a list of `range` (the reference's Vec<Range<NI>>)"""
        return [range(a, b) for a, b in degree_partition_of_offsets(self.csr_out.host()[0], concurrency, self.edge_count())]

    def in_degree_partition(self, concurrency: int):
        """InDegreePartitionOp (graph_ops.rs:41-50, 405-440) — the partition the multi-GPU PageRank shards its rows by"""
        return [range(a, b) for a, b in degree_partition_of_offsets(self.csr_inc.host()[0], concurrency, self.edge_count())]

    def to_undirected(self, layout=None):
        """ToUndirectedOp (crates/builder/src/graph_ops.rs:176-230, csr.rs:391-464): an Undirected
        build over this graph's out-edges."""
        layout = self.layout if layout is None else layout
        h = vp()
        check(lib().gm_csr_to_undirected(self.csr_out.handle, int(layout), C.byref(h)))
        ug = UndirectedCsrGraph(DeviceCsr(h), layout)
        ug._node_values = None if self._node_values is None else list(self._node_values)  # csr.rs:400
        return ug


class UndirectedCsrGraph(_GraphBase):
    """UndirectedCsrGraph = one symmetrised CSR (crates/builder/src/graph/csr.rs:658-732)."""

    def __init__(self, csr: DeviceCsr, layout=CsrLayout.Unsorted):
        self.csr = csr
        self._n = csr.n
        self.layout = layout

    def edge_count(self):
        return self.csr.m // 2  # csr.rs:687-689

    def degree(self, u):
        self._check_node(u)
        off = self.csr.host()[0]
        return int(off[u + 1] - off[u])

    def neighbors(self, u):
        self._check_node(u)
        off, tgt, _ = self.csr.host()
        return tgt[off[u]:off[u + 1]]

    def degree_partition(self, concurrency: int):
        """DegreePartitionOp (graph_ops.rs:17-26, 331-366): batch = ceil(2 edge_count / concurrency) — every edge counts at
        both of its ends"""
        return [range(a, b) for a, b in degree_partition_of_offsets(self.csr.host()[0], concurrency, 2 * self.edge_count())]

    def make_degree_ordered(self):
        """RelabelByDegreeOp (graph_ops.rs:240-253, 511-638); swaps the CSR in place and returns
        the old-id -> new-id map."""
        new_id = np.empty(self._n, np.uint32)
        h = vp()
        check(lib().gm_csr_relabel_by_degree(self.csr.handle, C.byref(h), _ptr(new_id) if self._n else None))
        self.csr = DeviceCsr(h)
        return new_id


# ------------------------------------------------------------------------------------------------
# inputs (crates/builder/src/input/{edgelist,graph500}.rs) and the builder
# ------------------------------------------------------------------------------------------------
class EdgeListInput:
    """`source target[ weight]` per line, \\n or \\r\\n (input/edgelist.rs:181-265)."""

    def __init__(self, weighted=False):
        self.weighted = weighted

    def read(self, path):
        cols = 3 if self.weighted else 2
        with open(path, "rb") as f:
            toks = f.read().split()
        if len(toks) % cols:
            raise ValueError(f"{path}: malformed edge list")
        arr = np.array(toks).reshape(-1, cols)
        src = arr[:, 0].astype(np.uint64)
        dst = arr[:, 1].astype(np.uint64)
        w = arr[:, 2].astype(np.float32) if self.weighted else None
        n = int(max(src.max(), dst.max())) + 1 if src.size else 0  # csr.rs:530: max id + 1
        return src, dst, w, n


class Graph500Input:
    """12-byte packed edges (input/graph500.rs:111-127); node_count = edge_count / 16 (:74)."""

    def read(self, path):
        raw = np.fromfile(path, dtype="<u4")
        if raw.size % 3:
            raise ValueError(f"{path}: size is not a multiple of 12 bytes")
        raw = raw.reshape(-1, 3)
        hi = raw[:, 2].astype(np.uint64)
        src = raw[:, 0].astype(np.uint64) | ((hi & np.uint64(0xFFFF)) << np.uint64(32))
        dst = raw[:, 1].astype(np.uint64) | ((hi >> np.uint64(16)) << np.uint64(32))
        return src, dst, None, raw.shape[0] // 16


class GraphBuilder:
    """GraphBuilder::new().csr_layout(..).edges(..) | .file_format(..).path(..) -> build()
    (crates/builder/src/builder.rs:123-540).  ``build(kind)`` takes the graph type the Rust
    caller would name in its type annotation."""

    def __init__(self):
        self._layout = CsrLayout.Unsorted
        self._edges = None
        self._format = None
        self._path = None
        self._device = 0
        self._node_values = None

    def csr_layout(self, layout):
        self._layout = CsrLayout(layout)
        return self

    def device(self, device):
        self._device = device
        return self

    def edges(self, edges):
        e = np.asarray(list(edges), dtype=np.uint64).reshape(-1, 2)
        self._edges = (e[:, 0], e[:, 1], None)
        return self

    def edges_with_values(self, edges):
        lst = list(edges)
        s = np.array([x[0] for x in lst], np.uint64)
        d = np.array([x[1] for x in lst], np.uint64)
        w = np.array([x[2] for x in lst], np.float32)
        self._edges = (s, d, w)
        return self

    def node_values(self, values):
        """builder.rs:388-403, 425-440 (after .edges / .edges_with_values): one value per node; their NUMBER is the graph's node
        count — it may exceed the edge list's largest id + 1 (csr.rs:1221-1261), it must not be smaller (csr.rs:552-563: panic)"""
        if self._edges is None:
            raise ValueError("GraphBuilder: node_values() follows edges() / edges_with_values()")  # (a type error in the reference)
        self._node_values = list(values)
        return self

    def file_format(self, fmt):
        self._format = fmt
        return self

    def path(self, path):
        self._path = os.fspath(path)
        return self

    def build(self, kind=None):
        kind = kind or DirectedCsrGraph
        if self._edges is not None:
            src, dst, w = self._edges
            n = int(max(src.max(), dst.max())) + 1 if src.size else 0
            if self._node_values is not None:
                if len(self._node_values) < n:
                    raise ValueError(f"number of node values ({len(self._node_values)}) does not match node count of edge list ({n})")
                n = len(self._node_values)
        elif self._format is not None and self._path is not None:
            src, dst, w, n = self._format.read(self._path)
        else:
            raise ValueError("GraphBuilder: no edges and no file_format/path")
        if n >= 2**32 or (src.size and max(int(src.max()), int(dst.max())) >= 2**32):
            raise OverflowError("node ids do not fit the u32 device id type")
        src, dst = src.astype(np.uint32), dst.astype(np.uint32)
        if kind is DirectedCsrGraph:
            out = DeviceCsr.from_edges(n, src, dst, w, Direction.Outgoing, self._layout, self._device)
            inc = DeviceCsr.from_edges(n, src, dst, w, Direction.Incoming, self._layout, self._device)
            g = DirectedCsrGraph(out, inc, self._layout)
            g._node_values = self._node_values
            return g
        if kind is UndirectedCsrGraph:
            csr = DeviceCsr.from_edges(n, src, dst, w, Direction.Undirected, self._layout, self._device)
            g = UndirectedCsrGraph(csr, self._layout)
            g._node_values = self._node_values
            return g
        raise TypeError(f"unknown graph kind {kind!r}")


# ------------------------------------------------------------------------------------------------
# algorithms
# ------------------------------------------------------------------------------------------------
@dataclass
class PageRankConfig:
    """crates/algos/src/page_rank.rs:14-56"""
    max_iterations: int = 20
    tolerance: float = 1e-4
    damping_factor: float = 0.85

    DEFAULT_MAX_ITERATIONS = 20
    DEFAULT_TOLERANCE = 1e-4
    DEFAULT_DAMPING_FACTOR = 0.85


class PageRankMode(enum.IntEnum):
    Auto = 0        # n <= 16384: Sequential (bit-exact with the reference), else Jacobi
    Jacobi = 1      # synchronous sweeps, engine picked by graph size
    Sequential = 2
    JacobiPull = 3  # force the pull-tile sweep kernels
    JacobiPB = 4    # force the propagation-blocking sweep kernels
    JacobiRefOrder = 5  # synchronous sweeps with the reference's left-to-right f32 row sums (parity instrument)
    BlockGS = 6     # block-Gauss-Seidel sweeps (the reference's in-place update at block granularity; PB engine)


def page_rank(graph: DirectedCsrGraph, config: PageRankConfig | None = None, mode=PageRankMode.Auto):
    """page_rank(&graph, config) -> (scores, iterations, error) — crates/algos/src/page_rank.rs:58-111."""
    config = config or PageRankConfig()
    n = graph.node_count()
    scores = _result_buffer(n, np.float32)
    it, err = u64(0), f64(0.0)
    # both CSRs are resident: out-degrees are taken from the out-CSR's offsets on the device
    check(lib().gm_page_rank_directed(graph.csr_out.handle, graph.csr_inc.handle, int(config.max_iterations),
                                      float(config.tolerance), float(config.damping_factor), int(mode),
                                      _ptr(scores) if n else None, C.byref(it), C.byref(err)))
    return scores, int(it.value), float(err.value)


def page_rank_multi(graph: DirectedCsrGraph, config: PageRankConfig | None = None, devices=None, n_devices: int | None = None):
    """page_rank over several GPUs of one node through the C ABI (gm_page_rank_multi: 1-D in-degree-balanced row
    ranges, RCCL all-gather of out_scores per sweep, one host thread).  devices: list of device ordinals (a device
    named twice = virtual ranks on one GPU), or n_devices for 0 .. n_devices-1."""
    config = config or PageRankConfig()
    n = graph.node_count()
    scores = _result_buffer(n, np.float32)
    it, err = u64(0), f64(0.0)
    if devices is not None:
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        count = len(devices)
    else:
        arr, count = None, int(n_devices or 1)
    check(lib().gm_page_rank_multi(graph.csr_out.handle, graph.csr_inc.handle, arr, count, int(config.max_iterations),
                                   float(config.tolerance), float(config.damping_factor), _ptr(scores) if n else None,
                                   C.byref(it), C.byref(err)))
    return scores, int(it.value), float(err.value)


def page_rank_multi_slices(slices, bounds, out_degree_full, config: PageRankConfig | None = None, devices=None):
    """gm_page_rank_multi_slices: the partitioned page_rank from PIECES — slices[p]: DeviceCsr with the rows [bounds[p],
    bounds[p + 1]) of the in-CSR on devices[p] (targets: global node ids); out_degree_full[p]: device address of u32[n] on
    devices[p] (or an object with data_ptr()).  No device holds the whole graph (graph_amd/distributed.py:
    partition_local_slices builds such pieces)."""
    config = config or PageRankConfig()
    count = len(slices)
    n = int(bounds[-1])
    devices = list(devices) if devices is not None else list(range(count))
    scores = _result_buffer(n, np.float32)
    it, err = u64(0), f64(0.0)
    hs = (vp * count)(*[s.handle for s in slices])
    bd = (C.c_uint64 * (count + 1))(*[int(b) for b in bounds])
    od = (C.c_uint64 * count)(*[int(o.data_ptr() if hasattr(o, "data_ptr") else o) for o in out_degree_full])
    dv = (C.c_int * count)(*[int(d) for d in devices])
    check(lib().gm_page_rank_multi_slices(hs, bd, od, n, dv, count, int(config.max_iterations), float(config.tolerance),
                                          float(config.damping_factor), _ptr(scores) if n else None, C.byref(it), C.byref(err)))
    return scores, int(it.value), float(err.value)


@dataclass
class WccConfig:
    """crates/algos/src/wcc.rs:43-79 (chunk_size is a CPU scheduling knob; ignored on the device)"""
    chunk_size: int = 16384
    neighbor_rounds: int = 2
    sampling_size: int = 1024


class Components:
    """Components<NI> (wcc.rs:95-99)"""

    def __init__(self, labels):
        self._labels = labels

    def component(self, node):
        return int(self._labels[node])

    def to_vec(self):
        return self._labels


def wcc_afforest(graph: DirectedCsrGraph, config: WccConfig | None = None, device_out=None) -> Components:
    """wcc_afforest — crates/algos/src/wcc.rs:127-141.
    device_out (optional): an object with data_ptr() over u32 / i32 [n] on the graph's device (e.g. a torch tensor) — the
    component ids are left THERE and it is returned instead of a Components: a caller that goes on working on the GPU
    spares the n * 4 bytes over PCIe (16.8 MB = 0.3 ms of a 0.6 ms call at RMAT scale 22)."""
    config = config or WccConfig()
    if device_out is not None:
        check(lib().gm_wcc_afforest(graph.csr_out.handle, graph.csr_inc.handle, int(config.neighbor_rounds),
                                    int(config.sampling_size), C.c_void_p(int(device_out.data_ptr()))))
        return device_out
    labels = _result_buffer(graph.node_count(), np.uint32)
    check(lib().gm_wcc_afforest(graph.csr_out.handle, graph.csr_inc.handle, int(config.neighbor_rounds),
                                int(config.sampling_size), _ptr(labels) if labels.size else None))
    return Components(labels)


def wcc_afforest_dss(graph: DirectedCsrGraph, config: WccConfig | None = None) -> Components:
    """wcc_afforest_dss — crates/algos/src/wcc.rs:144-156.  The union-find backend is a CPU data
    structure choice; component(u) (the root = minimum id) is identical, so the device path is shared."""
    return wcc_afforest(graph, config)


def wcc_baseline(graph: DirectedCsrGraph, config: WccConfig | None = None) -> Components:
    """wcc_baseline — crates/algos/src/wcc.rs:103-122"""
    labels = _result_buffer(graph.node_count(), np.uint32)
    check(lib().gm_wcc_baseline(graph.csr_out.handle, _ptr(labels) if labels.size else None))
    return Components(labels)


@dataclass
class DeltaSteppingConfig:
    """crates/algos/src/sssp.rs:18-36"""
    start_node: int
    delta: float


def delta_stepping(graph: DirectedCsrGraph, config: DeltaSteppingConfig, device_out=None):
    """delta_stepping — crates/algos/src/sssp.rs:38-102; returns f32 distances, f32::MAX = unreachable.
    device_out (optional): an object with data_ptr() over f32[n] on the graph's device (e.g. a torch tensor) — the distances
    are left THERE and it is returned: a caller that goes on working on the GPU spares the n * 4 bytes over PCIe (64 MB =
    1.2 ms of a 6.5 ms call at RMAT scale 24)."""
    if not 0 <= config.start_node < graph.node_count():
        raise IndexError(f"start_node {config.start_node} out of range")  # sssp.rs:52 panics
    if device_out is not None:
        check(lib().gm_sssp_delta_stepping(graph.csr_out.handle, int(config.start_node), float(config.delta),
                                           C.c_void_p(int(device_out.data_ptr()))))
        return device_out
    dist = _result_buffer(graph.node_count(), np.float32)
    check(lib().gm_sssp_delta_stepping(graph.csr_out.handle, int(config.start_node), float(config.delta),
                                       _ptr(dist)))
    return dist


def global_triangle_count(graph: UndirectedCsrGraph) -> int:
    """global_triangle_count — crates/algos/src/triangle_count.rs:22-86"""
    out = u64(0)
    check(lib().gm_triangle_count(graph.csr.handle, C.byref(out)))
    return int(out.value)


def relabel_graph(graph: UndirectedCsrGraph):
    """relabel_graph — crates/algos/src/triangle_count.rs:12-20"""
    graph.make_degree_ordered()


__all__ = [
    "CsrLayout", "Direction", "DeviceCsr", "DirectedCsrGraph", "UndirectedCsrGraph", "EdgeListInput",
    "Graph500Input", "GraphBuilder", "PageRankConfig", "PageRankMode", "page_rank", "page_rank_multi", "trim_device", "arena_info", "WccConfig", "Components",
    "wcc_afforest", "wcc_afforest_dss", "wcc_baseline", "DeltaSteppingConfig", "delta_stepping",
    "global_triangle_count", "relabel_graph",
]
