"""Bad inputs must come back as a status, never as out-of-bounds device work: endpoints >= n in the
device CSR build, corrupt uploaded / wrapped CSR arrays, truncated dumps; plus the put-back semantics of a
hand-made list with a single self-loop entry (reachable only through an upload)."""
import ctypes as C
import io as _io
import struct

import numpy as np
import pytest


@pytest.fixture(scope="module")
def P():
    from graph_amd import prelude

    return prelude


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("direction", [0, 1, 2])
def test_build_rejects_endpoint_beyond_node_count(P, layout, direction):
    from graph_amd._lib import GraphMI355XError

    rng = np.random.default_rng(5)
    n, m = 1000, 20000
    s = rng.integers(0, n, m).astype(np.uint32)
    d = rng.integers(0, n, m).astype(np.uint32)
    for arr, bad in ((s, n), (d, 0xFFFFFFF0), (s, 1 << 20)):
        keep = arr[1234]
        arr[1234] = bad  # n itself, a huge id, an id far beyond the sorted row bits
        with pytest.raises(GraphMI355XError) as e:
            P.DeviceCsr.from_edges(n, s, d, None, P.Direction(direction), P.CsrLayout(layout))
        assert "node_count" in str(e.value)
        arr[1234] = keep
    # the device is still healthy and the same call works on clean input
    g = P.DeviceCsr.from_edges(n, s, d, None, P.Direction(direction), P.CsrLayout(layout))
    assert g.n == n
    # Graph500Input sets n = edges / 16 (0 below 16 edges): every id is then out of range -> an error, not a hang
    with pytest.raises(GraphMI355XError):
        P.DeviceCsr.from_edges(0, s[:8], d[:8], None, P.Direction(direction), P.CsrLayout(layout))


@pytest.mark.gpu
def test_upload_and_wrap_validate_the_arrays(P):
    import torch
    from graph_amd._lib import GraphMI355XError, check, lib, vp

    off = np.array([0, 2, 3, 5], np.uint32)
    tgt = np.array([1, 2, 0, 0, 1], np.uint32)
    assert P.DeviceCsr.from_arrays(off, tgt).m == 5
    with pytest.raises(GraphMI355XError):  # a target >= n
        P.DeviceCsr.from_arrays(off, np.array([1, 2, 0, 3, 1], np.uint32))
    with pytest.raises(GraphMI355XError):  # offsets not ascending
        P.DeviceCsr.from_arrays(np.array([0, 3, 2, 5], np.uint32), tgt)
    t_off = torch.tensor([0, 2, 3, 5], dtype=torch.int32, device="cuda")
    t_bad = torch.tensor([1, 2, 0, 7, 1], dtype=torch.int32, device="cuda")
    h = vp()
    with pytest.raises(GraphMI355XError):
        check(lib().gm_csr_wrap_device(t_off.data_ptr(), t_bad.data_ptr(), 0, 3, 5, 0, C.byref(h)))


def test_truncated_dump_is_an_error_not_a_short_array():
    from graph_amd import io

    buf = _io.BytesIO()
    io._write_csr(buf, np.array([0, 2, 3], np.uint32), np.array([1, 1, 0], np.uint32), None)
    raw = buf.getvalue()
    for cut in (4, 10, 15, 25, len(raw) - 1):
        with pytest.raises(ValueError):
            io._read_csr(_io.BytesIO(raw[:cut]), weighted=False)
    with pytest.raises(ValueError):  # absurd type-name length
        io._read_csr(_io.BytesIO(struct.pack("<Q", 1 << 40) + b"u32"), weighted=False)


@pytest.mark.gpu
def test_triangle_count_single_self_loop_entry(P, oracle):
    # N(0) = [0, 1], N(1) = [0]: strictly increasing lists WITH a self-loop listed once (an undirected build
    # would list it twice).  The put-back walk counts (u, v, w) = (0, 0, 0) and (1, 0, 0): 2.
    off = np.array([0, 2, 3], np.uint32)
    tgt = np.array([0, 1, 0], np.uint32)
    expect = oracle.triangle_count(off, tgt)
    assert expect == 2
    ug = P.UndirectedCsrGraph(P.DeviceCsr.from_arrays(off, tgt), P.CsrLayout.Sorted)
    assert P.global_triangle_count(ug) == expect
    # a larger random set-valued graph with a few single self-loops, against the oracle
    rng = np.random.default_rng(11)
    n = 600
    adj = [set() for _ in range(n)]
    for a, b in rng.integers(0, n, (6000, 2)):
        if a != b:
            adj[a].add(int(b))
            adj[b].add(int(a))
    for u in rng.integers(0, n, 25):
        adj[u].add(int(u))
    lists = [sorted(x) for x in adj]
    off = np.zeros(n + 1, np.uint32)
    off[1:] = np.cumsum([len(x) for x in lists])
    tgt = np.array([v for x in lists for v in x], np.uint32)
    ug = P.UndirectedCsrGraph(P.DeviceCsr.from_arrays(off, tgt), P.CsrLayout.Sorted)
    assert P.global_triangle_count(ug) == oracle.triangle_count(off, tgt)
