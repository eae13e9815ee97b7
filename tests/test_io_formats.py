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


def test_cpp_prelude_file_readers_match_python_readers(golden_dir):
    """include/graph_prelude.hpp's EdgeListInput / Graph500Input (host-only parsing) against the Python
    readers on the reference's fixture files; the program makes no library call, so this runs without a GPU."""
    import subprocess

    from graph_amd import prelude as P

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tests", "cpp"), "loaders_test"])
    out = subprocess.run([os.path.join(root, "tests", "cpp", "loaders_test"), golden_dir], capture_output=True, text=True,
                         check=True).stdout.strip().splitlines()
    got = {ln.split()[0]: dict(kv.split("=") for kv in ln.split()[1:]) for ln in out if " " in ln}
    assert out[-1] == "missing-file-throws=1"
    # the header's greedy_node_map_partition on the reference's own unit tests (crates/builder/src/graph_ops.rs:673-708)
    assert got["partition_1_part"]["ranges"] == "0-10"
    assert got["partition_2_parts"]["ranges"] == "0-8,8-10"
    assert got["partition_6_parts"]["ranges"] == "0-4,4-6,6-7,7-8,8-9,9-10"
    assert got["partition_max_batches"]["ranges"] == "0-4,4-6,6-10"
    assert got["partition_empty"]["ranges"] == ""

    def fnv(src, dst):
        h = 1469598103934665603
        for s, d in zip(src.tolist(), dst.tolist()):
            h = ((h ^ s) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            h = ((h ^ d) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    cases = {"example.el": P.EdgeListInput(), "example.wel": P.EdgeListInput(weighted=True),
             "windows.el": P.EdgeListInput(), "scale_8.graph500": P.Graph500Input()}
    for name, reader in cases.items():
        src, dst, w, n = reader.read(os.path.join(golden_dir, name))
        rec = got[name]
        assert int(rec["nodes"]) == n and int(rec["edges"]) == src.size
        assert int(rec["hash"]) == fnv(src, dst)
        assert int(rec["values"]) == (0 if w is None else w.size)
        if w is not None:
            assert abs(float(rec["wsum"]) - float(w.astype(np.float64).sum())) < 1e-6
