"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/graph_mi355x.h declares; host logic behaves; calls fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "graph_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from graph_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/graph_mi355x.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared
    assert L.gm_abi_version() == 1


def test_calls_fail_loudly_without_a_gpu():
    import graph_amd
    from graph_amd import prelude as P

    if graph_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(graph_amd.GraphMI355XError) as ei:
        P.GraphBuilder().edges([(0, 1)]).build(P.DirectedCsrGraph)
    assert ei.value.status in (-1, -3)


def test_argument_validation_needs_no_gpu():
    from graph_amd import _lib

    L = _lib.lib()
    out = C.c_void_p()
    off = np.array([0, 1], np.uint32)
    tgt = np.array([0], np.uint32)
    assert L.gm_csr_upload_u32(None, None, None, 1, 1, 0, C.byref(out)) == _lib.GM_ERR_INVALID
    assert b"null" in L.gm_last_error()
    assert L.gm_csr_upload_u32(off.ctypes.data_as(C.c_void_p), tgt.ctypes.data_as(C.c_void_p), None, 1 << 33, 1, 0,
                               C.byref(out)) == _lib.GM_ERR_RANGE
    bad = np.array([1, 1], np.uint32)
    assert L.gm_csr_upload_u32(bad.ctypes.data_as(C.c_void_p), tgt.ctypes.data_as(C.c_void_p), None, 1, 1, 0,
                               C.byref(out)) == _lib.GM_ERR_INVALID
    it, err = C.c_uint64(), C.c_double()
    assert L.gm_page_rank(None, None, 20, 1e-4, 0.85, 0, None, C.byref(it), C.byref(err)) == _lib.GM_ERR_INVALID
    assert L.gm_triangle_count(None, None) == _lib.GM_ERR_INVALID
    assert L.gm_sssp_delta_stepping(None, 0, 1.0, None) == _lib.GM_ERR_INVALID
    assert L.gm_csr_node_count(None) == 0


def test_input_readers(golden_dir):
    from graph_amd import prelude as P

    s, d, w, n = P.EdgeListInput().read(os.path.join(golden_dir, "test.el"))
    assert n == 5 and list(zip(s.tolist(), d.tolist())) == [(0, 1), (0, 2), (1, 2), (1, 3), (2, 4), (3, 4)]
    s, d, w, n = P.EdgeListInput(weighted=True).read(os.path.join(golden_dir, "test.wel"))
    assert n == 5 and np.allclose(w, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    s, d, w, n = P.EdgeListInput().read(os.path.join(golden_dir, "windows.el"))
    assert n == 4 and s.size == 3
    s, d, w, n = P.Graph500Input().read(os.path.join(golden_dir, "scale_8.graph500"))
    assert n == 256 and s.size == 4096
    assert list(zip(s[:5].tolist(), d[:5].tolist())) == [(17, 138), (82, 127), (250, 60), (57, 6), (46, 206)]


def test_config_defaults():
    from graph_amd import prelude as P

    c = P.PageRankConfig()
    assert (c.max_iterations, c.tolerance, c.damping_factor) == (20, 1e-4, 0.85)  # page_rank.rs:46-48
    w = P.WccConfig()
    assert (w.chunk_size, w.neighbor_rounds, w.sampling_size) == (16384, 2, 1024)  # wcc.rs:69-71
    assert int(P.CsrLayout.Unsorted) == 0 and int(P.CsrLayout.Deduplicated) == 2


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: the header must compile as C (what a cgo / bindgen / ctypes user feeds it to)."""
    import subprocess

    src = tmp_path / "hc.c"
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "graph_mi355x.h")
    src.write_text('#include "%s"\nint main(void) { return gm_abi_version() > 0 ? 0 : 1; }\n' % header)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
