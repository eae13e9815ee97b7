"""On-disk formats: Graph500 packed edges and the binary CSR dump (graph_amd/io.py)."""
import io as _io
import os
import struct

import numpy as np
import pytest


def test_graph500_writer_round_trips_the_reference_fixture(tmp_path, golden_dir):
    from graph_amd import io, prelude as P

    src, dst, _, n = P.Graph500Input().read(os.path.join(golden_dir, "scale_8.graph500"))
    out = tmp_path / "copy.graph500"
    io.write_graph500(out, src, dst)
    assert open(out, "rb").read() == open(os.path.join(golden_dir, "scale_8.graph500"), "rb").read()
    big_s, big_d = np.array([1 << 40, 5], np.uint64), np.array([7, (1 << 47) + 3], np.uint64)
    io.write_graph500(out, big_s, big_d)
    s2, d2, _, _ = P.Graph500Input().read(out)
    assert np.array_equal(s2, big_s) and np.array_equal(d2, big_d)
    with pytest.raises(OverflowError):
        io.write_graph500(out, np.array([1 << 48], np.uint64), np.array([0], np.uint64))


def test_binary_csr_layout_bytes():
    from graph_amd import io

    buf = _io.BytesIO()
    io._write_csr(buf, np.array([0, 2, 3], np.uint32), np.array([1, 1, 0], np.uint32), None)
    raw = buf.getvalue()
    assert raw[:8] == struct.pack("<Q", 3) and raw[8:11] == b"u32"
    assert struct.unpack("<II", raw[11:19]) == (2, 3)
    assert np.array_equal(np.frombuffer(raw[19:31], "<u4"), [0, 2, 3])
    assert np.array_equal(np.frombuffer(raw[31:], "<u4"), [1, 1, 0])
    buf = _io.BytesIO()
    io._write_csr(buf, np.array([0, 1], np.uint32), np.array([0], np.uint32), np.array([0.5], np.float32))
    assert len(buf.getvalue()) == 8 + 3 + 8 + 8 + 8  # Target<u32, f32> is an 8-byte AoS record
    buf.seek(0)
    off, tgt, w = io._read_csr(buf, weighted=True)
    assert list(off) == [0, 1] and list(tgt) == [0] and list(w) == [0.5]
    bad = _io.BytesIO(struct.pack("<Q", 3) + b"u64" + struct.pack("<II", 0, 0) + b"\0" * 4)
    with pytest.raises(ValueError):
        io._read_csr(bad, weighted=False)


@pytest.mark.gpu
def test_serialize_round_trip_on_device(tmp_path, golden_dir):
    from graph_amd import io, prelude as P

    g = (P.GraphBuilder().csr_layout(P.CsrLayout.Sorted).file_format(P.Graph500Input())
         .path(os.path.join(golden_dir, "scale_8.graph500")).build(P.DirectedCsrGraph))
    path = tmp_path / "scale8.bin"
    io.serialize(g, path)
    # NodeValues<()> count (8) + 2 x (8 + 3 + 8 + 257*4 + 4096*4)
    assert os.path.getsize(path) == 8 + 2 * (8 + 3 + 8 + 257 * 4 + 4096 * 4)
    h = io.deserialize(path, P.DirectedCsrGraph, layout=P.CsrLayout.Sorted)
    assert h.node_count() == 256 and h.edge_count() == 4096
    assert list(h.in_neighbors(0)) == list(g.in_neighbors(0)) and list(h.out_neighbors(0)) == [37, 157]
    a, b = P.page_rank(g), P.page_rank(h)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    ug = g.to_undirected(P.CsrLayout.Deduplicated)
    io.serialize(ug, path)
    uh = io.deserialize(path, P.UndirectedCsrGraph, layout=P.CsrLayout.Deduplicated)
    assert P.global_triangle_count(uh) == 10508
