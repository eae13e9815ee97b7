"""On-disk formats of the reference, wire-compatible:

* Graph500 packed edges, 12 bytes each: v0_low u32, v1_low u32, high u32 with
  source = v0_low | (high & 0xFFFF) << 32, target = v1_low | (high >> 16) << 32
  (crates/builder/src/input/graph500.rs:111-127); node_count = edge_count / 16 (:74).
* Binary CSR dump of SerializeGraphOp (crates/builder/src/graph/csr.rs:247-362, 606-656):
  node values block = usize count + raw values (NV = (): no bytes), then per CSR: usize length of the
  id type name + the name ("u32") + [node_count, edge_count] as NI + raw offsets + raw targets
  (Target<u32, ()> = 4 bytes, Target<u32, f32> = {u32, f32} AoS).  Directed = out CSR then in CSR.

Pure host byte formats (numpy); the graphs they feed live on the device via graph_amd.prelude.
"""
from __future__ import annotations

import struct

import numpy as np

from . import prelude as P

_TYPE_NAME = b"u32"  # std::any::type_name::<u32>()


def write_graph500(path, src, dst):
    src = np.asarray(src, np.uint64)
    dst = np.asarray(dst, np.uint64)
    if src.size and (int(src.max()) >= 1 << 48 or int(dst.max()) >= 1 << 48):
        raise OverflowError("Graph500 packed edges hold 48-bit ids")
    rec = np.empty((src.size, 3), "<u4")
    rec[:, 0] = (src & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    rec[:, 1] = (dst & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    rec[:, 2] = ((src >> np.uint64(32)) | ((dst >> np.uint64(32)) << np.uint64(16))).astype(np.uint32)
    rec.tofile(path)


def _write_csr(f, offsets, targets, weights):
    f.write(struct.pack("<Q", len(_TYPE_NAME)))
    f.write(_TYPE_NAME)
    n, m = offsets.size - 1, targets.size
    f.write(struct.pack("<II", n, m))
    f.write(np.ascontiguousarray(offsets, "<u4").tobytes())
    if weights is None:
        f.write(np.ascontiguousarray(targets, "<u4").tobytes())
    else:
        rec = np.empty(m, dtype=[("target", "<u4"), ("value", "<f4")])
        rec["target"], rec["value"] = targets, weights
        f.write(rec.tobytes())


def _read_exact(f, count, what):
    """a truncated dump must fail here, not as a short array handed to the device"""
    buf = f.read(count)
    if len(buf) != count:
        raise ValueError(f"truncated graph file: {what} needs {count} bytes, {len(buf)} left")
    return buf


def _read_csr(f, weighted):
    (name_len,) = struct.unpack("<Q", _read_exact(f, 8, "type-name length"))
    if name_len > 64:
        raise ValueError(f"invalid id type: a type name of {name_len} bytes")
    name = _read_exact(f, name_len, "type name")
    if name != _TYPE_NAME:  # Error::InvalidIdType (csr.rs:284-289)
        raise ValueError(f"invalid id type: expected {_TYPE_NAME.decode()}, got {name.decode(errors='replace')}")
    n, m = struct.unpack("<II", _read_exact(f, 8, "node and edge count"))
    offsets = np.frombuffer(_read_exact(f, 4 * (n + 1), "offsets"), "<u4").copy()
    if weighted:
        rec = np.frombuffer(_read_exact(f, 8 * m, "targets"), dtype=[("target", "<u4"), ("value", "<f4")])
        return offsets, rec["target"].copy(), rec["value"].copy()
    return offsets, np.frombuffer(_read_exact(f, 4 * m, "targets"), "<u4").copy(), None


def serialize(graph, path):
    """SerializeGraphOp::serialize for DirectedCsrGraph<u32> / UndirectedCsrGraph<u32> (NV = ())."""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", graph.node_count()))  # NodeValues<()>: count, no payload
        if isinstance(graph, P.DirectedCsrGraph):
            for csr in (graph.csr_out, graph.csr_inc):
                _write_csr(f, *csr.host())
        else:
            _write_csr(f, *graph.csr.host())


def deserialize(path, kind=P.DirectedCsrGraph, weighted=False, layout=P.CsrLayout.Unsorted, device=0):
    """DeserializeGraphOp::deserialize; uploads the CSR(s) to the device."""
    with open(path, "rb") as f:
        (node_values,) = struct.unpack("<Q", _read_exact(f, 8, "node-value count"))
        if kind is P.DirectedCsrGraph:
            out = P.DeviceCsr.from_arrays(*_read_csr(f, weighted), device=device)
            inc = P.DeviceCsr.from_arrays(*_read_csr(f, weighted), device=device)
            assert out.n == node_values
            return P.DirectedCsrGraph(out, inc, layout)
        csr = P.DeviceCsr.from_arrays(*_read_csr(f, weighted), device=device)
        return P.UndirectedCsrGraph(csr, layout)
