"""ctypes/numpy front for oracle/liborc.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package graph_amd never does.  Every function is a thin typed wrapper over the C
restatement in graph_oracle.c, which cites the reference file:line it follows.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")

OUTGOING, INCOMING, UNDIRECTED = 0, 1, 2
UNSORTED, SORTED, DEDUPLICATED = 0, 1, 2
AFFOREST, AFFOREST_DSS, BASELINE = 0, 1, 2

_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "graph_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_native = None
_native_flags = None


def native_lib():
    """The build used by the TIMED cpu_baseline legs only: `-O3 -march=native`, compiled on the
    box whose cores are timed (oracle/Makefile: liborc_native.so; never shipped).  Falls back to liborc.so
    (-O3, generic x86-64) when no compiler is there; timed_build_flags() says which one ran."""
    global _native, _native_flags
    if _native is None:
        path = os.path.join(_HERE, "liborc_native.so")
        try:
            subprocess.check_call(["make", "-C", _HERE, "-B", "liborc_native.so"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            _native = _bind(C.CDLL(path))
            _native_flags = "gcc -O3 -march=native -ffp-contract=off"
        except (OSError, subprocess.CalledProcessError):
            _native = lib()
            _native_flags = "gcc -O3 -ffp-contract=off (generic x86-64: native build failed)"
    return _native


def timed_build_flags() -> str:
    native_lib()
    return _native_flags


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    L.orc_rmat_edges.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, _u32p, _u32p]
    L.orc_rmat_edges.restype = None
    L.orc_rmat_weights.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
    L.orc_rmat_weights.restype = None
    L.orc_csr_build.argtypes = [C.c_uint32, C.c_uint64, _u32p, _u32p, C.c_void_p, C.c_int, C.c_int,
                                _u32p, _u32p, C.c_void_p]
    L.orc_csr_build.restype = C.c_uint64
    L.orc_relabel_by_degree.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p]
    L.orc_relabel_by_degree.restype = C.c_int
    L.orc_page_rank_seq.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint64, C.c_double, C.c_float,
                                    _f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.orc_page_rank_seq.restype = None
    L.orc_page_rank_chunked.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint64, C.c_double, C.c_float,
                                        C.c_uint32, _f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.orc_page_rank_chunked.restype = C.c_int
    L.orc_page_rank_chunked_timed.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint64, C.c_float, C.c_uint32, C.c_int,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_page_rank_chunked_timed.restype = C.c_int
    L.orc_page_rank_f64.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint64, C.c_double, C.c_double,
                                    _f64p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.orc_page_rank_f64.restype = None
    L.orc_page_rank_jacobi_sweep.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_float, _f32p, _f32p, _f32p]
    L.orc_page_rank_jacobi_sweep.restype = C.c_double
    L.orc_uf_new.argtypes = [C.c_uint32, _u32p]
    L.orc_uf_union.argtypes = [C.c_int, _u32p, C.c_uint32, C.c_uint32]
    L.orc_uf_find.argtypes = [C.c_int, _u32p, C.c_uint32]
    L.orc_uf_find.restype = C.c_uint32
    L.orc_uf_compress.argtypes = [C.c_int, _u32p, C.c_uint32]
    L.orc_wcc.argtypes = [C.c_int, C.c_uint32, _u32p, _u32p, _u32p, _u32p, C.c_uint64, C.c_uint64,
                          C.c_uint64, _u32p]
    L.orc_wcc.restype = C.c_int
    L.orc_delta_stepping.argtypes = [C.c_uint32, _u32p, _u32p, _f32p, C.c_uint64, C.c_float, _f32p]
    L.orc_delta_stepping.restype = C.c_int
    L.orc_wcc_afforest_timed.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_uint32, _u32p, C.POINTER(C.c_double)]
    L.orc_wcc_afforest_timed.restype = C.c_int
    L.orc_delta_stepping_timed.argtypes = [C.c_uint32, _u32p, _u32p, _f32p, C.c_uint64, C.c_float, C.c_uint32, _f32p,
                                           C.POINTER(C.c_double)]
    L.orc_delta_stepping_timed.restype = C.c_int
    L.orc_triangle_count.argtypes = [C.c_uint32, _u32p, _u32p, C.c_uint32]
    L.orc_triangle_count.restype = C.c_uint64
    L.orc_greedy_degree_partition.argtypes = [C.c_uint32, _u32p, C.c_uint32, _u32p]
    L.orc_greedy_degree_partition.restype = C.c_uint32
    return L


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------
def rmat_edges(scale: int, seed: int = 42, edge_factor: int = 16, first: int = 0, count: int | None = None):
    m = (edge_factor << scale) if count is None else count
    src = np.empty(m, np.uint32)
    dst = np.empty(m, np.uint32)
    lib().orc_rmat_edges(scale, seed, first, m, src, dst)
    return src, dst


def rmat_weights(m: int, seed: int = 44, first: int = 0):
    w = np.empty(m, np.float32)
    lib().orc_rmat_weights(seed, first, m, w)
    return w


def read_edge_list(path: str, weighted: bool = False):
    """`s t[ w]` per line, \\n or \\r\\n (crates/builder/src/input/edgelist.rs:181-265)."""
    src, dst, w = [], [], []
    with open(path, "rb") as f:
        for line in f.read().splitlines():
            parts = line.split()
            if not parts:
                continue
            src.append(int(parts[0]))
            dst.append(int(parts[1]))
            if weighted:
                w.append(float(parts[2]))
    s = np.asarray(src, np.uint32)
    d = np.asarray(dst, np.uint32)
    return (s, d, np.asarray(w, np.float32)) if weighted else (s, d)


def read_graph500(path: str):
    """12-byte packed edges (crates/builder/src/input/graph500.rs:111-127); n = edges/16 (:74)."""
    raw = np.fromfile(path, dtype=np.uint32).reshape(-1, 3)
    hi = raw[:, 2].astype(np.uint64)
    src = raw[:, 0].astype(np.uint64) | ((hi & 0xFFFF) << 32)
    dst = raw[:, 1].astype(np.uint64) | ((hi >> 16) << 32)
    n = raw.shape[0] // 16
    return src.astype(np.uint32), dst.astype(np.uint32), n


def csr_build(n: int, src, dst, direction: int, layout: int, weights=None):
    src = np.ascontiguousarray(src, np.uint32)
    dst = np.ascontiguousarray(dst, np.uint32)
    m = src.size
    cap = 2 * m if direction == UNDIRECTED else m
    off = np.empty(n + 1, np.uint32)
    tgt = np.empty(max(cap, 1), np.uint32)
    wout = None
    wp = None
    wo = None
    if weights is not None:
        weights = np.ascontiguousarray(weights, np.float32)
        wout = np.empty(max(cap, 1), np.float32)
        wp = weights.ctypes.data_as(C.c_void_p)
        wo = wout.ctypes.data_as(C.c_void_p)
    total = lib().orc_csr_build(n, m, src, dst, wp, direction, layout, off, tgt, wo)
    assert total != 2**64 - 1
    if weights is not None:
        return off, tgt[:total].copy(), wout[:total].copy()
    return off, tgt[:total].copy()


def relabel_by_degree(off, tgt):
    n = off.size - 1
    noff = np.empty_like(off)
    ntgt = np.empty(max(tgt.size, 1), np.uint32)
    new_id = np.empty(max(n, 1), np.uint32)
    rc = lib().orc_relabel_by_degree(n, off, np.ascontiguousarray(tgt), noff, ntgt, new_id)
    assert rc == 0
    return noff, ntgt[: tgt.size], new_id[:n]


def out_degrees_from(n: int, src) -> np.ndarray:
    return np.bincount(np.asarray(src, np.int64), minlength=n).astype(np.uint32)


# ----------------------------------------------------------------------------------------------
# algorithms
# ----------------------------------------------------------------------------------------------
def _tgt(a):
    a = np.ascontiguousarray(a, np.uint32)
    return a if a.size else np.zeros(1, np.uint32)


def page_rank_seq(in_off, in_tgt, out_deg, max_iterations=20, tolerance=1e-4, damping=0.85):
    n = in_off.size - 1
    scores = np.empty(max(n, 1), np.float32)
    it, err = C.c_uint64(), C.c_double()
    lib().orc_page_rank_seq(n, in_off, _tgt(in_tgt), np.ascontiguousarray(out_deg, np.uint32), max_iterations,
                            tolerance, damping, scores, C.byref(it), C.byref(err))
    return scores[:n], it.value, err.value


def effective_cores() -> int:
    """What Rust's available_parallelism() (page_rank.rs:128) would report: the affinity mask capped by the
    cgroup CPU quota (a container with cpu.max = 16 CPUs on a 256-thread host runs 16 workers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def page_rank_chunked(in_off, in_tgt, out_deg, max_iterations=20, tolerance=1e-4, damping=0.85, threads=0):
    n = in_off.size - 1
    threads = threads or effective_cores()
    scores = np.empty(max(n, 1), np.float32)
    it, err = C.c_uint64(), C.c_double()
    rc = lib().orc_page_rank_chunked(n, in_off, _tgt(in_tgt), np.ascontiguousarray(out_deg, np.uint32),
                                     max_iterations, tolerance, damping, threads, scores, C.byref(it),
                                     C.byref(err))
    assert rc == 0
    return scores[:n], it.value, err.value


def page_rank_chunked_timed(in_off, in_tgt, out_deg, sweeps, damping=0.85, threads=0, spread=True):
    """bench.py's CPU baseline: seconds for `sweeps` sweeps of the threaded path after one untimed sweep, on
    copies whose pages were first touched by all threads (spread) -> (seconds, last error)"""
    n = in_off.size - 1
    threads = threads or effective_cores()
    sec, err = C.c_double(), C.c_double()
    rc = native_lib().orc_page_rank_chunked_timed(n, in_off, _tgt(in_tgt), np.ascontiguousarray(out_deg, np.uint32), sweeps,
                                           damping, threads, 1 if spread else 0, C.byref(sec), C.byref(err))
    assert rc == 0
    return sec.value, err.value


def page_rank_f64(in_off, in_tgt, out_deg, max_iterations=500, tolerance=1e-14, damping=0.85):
    n = in_off.size - 1
    scores = np.empty(max(n, 1), np.float64)
    it, err = C.c_uint64(), C.c_double()
    lib().orc_page_rank_f64(n, in_off, _tgt(in_tgt), np.ascontiguousarray(out_deg, np.uint32), max_iterations,
                            tolerance, damping, scores, C.byref(it), C.byref(err))
    return scores[:n], it.value, err.value


def page_rank_jacobi_sweep(in_off, in_tgt, out_deg, damping, scores, outs_in):
    n = in_off.size - 1
    outs_out = np.empty_like(outs_in)
    err = lib().orc_page_rank_jacobi_sweep(n, in_off, _tgt(in_tgt), np.ascontiguousarray(out_deg, np.uint32),
                                           damping, scores, outs_in, outs_out)
    return outs_out, err


def wcc(out_off, out_tgt, in_off, in_tgt, algo=AFFOREST, neighbor_rounds=2, sampling_size=1024, seed=1, native=False):
    """native=True: the -march=native build (timed cpu_baseline legs; same results)"""
    n = out_off.size - 1
    comp = np.empty(max(n, 1), np.uint32)
    rc = (native_lib() if native else lib()).orc_wcc(algo, n, out_off, _tgt(out_tgt), in_off, _tgt(in_tgt), neighbor_rounds, sampling_size,
                       seed, comp)
    if rc != 0:
        raise ValueError(f"orc_wcc rc={rc}")
    return comp[:n]


def wcc_afforest_timed(out_off, out_tgt, in_off, in_tgt, threads=0, neighbor_rounds=2, sampling_size=1024, seed=1,
                       chunk_size=16384, native=True):
    """(labels, seconds): wcc_afforest on `threads` threads as the reference runs it on rayon (wcc.rs:186-301,
    afforest.rs:22-53) — the TIMED cpu_baseline leg; the labels equal wcc(...)'s"""
    n = out_off.size - 1
    comp = np.empty(max(n, 1), np.uint32)
    secs = C.c_double(0)
    rc = (native_lib() if native else lib()).orc_wcc_afforest_timed(n, out_off, _tgt(out_tgt), in_off, _tgt(in_tgt), neighbor_rounds,
                                                                    sampling_size, seed, chunk_size, threads or effective_cores(), comp,
                                                                    C.byref(secs))
    if rc != 0:
        raise ValueError(f"orc_wcc_afforest_timed rc={rc}")
    return comp[:n], secs.value


def delta_stepping_timed(off, tgt, w, start_node: int, delta: float, threads=0, native=True):
    """(distances, seconds): delta_stepping on `threads` threads with thread-local bins (sssp.rs:64-204) — the TIMED leg"""
    n = off.size - 1
    dist = np.empty(max(n, 1), np.float32)
    wv = np.ascontiguousarray(w, np.float32)
    secs = C.c_double(0)
    rc = (native_lib() if native else lib()).orc_delta_stepping_timed(n, off, _tgt(tgt), wv if wv.size else np.zeros(1, np.float32),
                                                                      start_node, delta, threads or effective_cores(), dist, C.byref(secs))
    if rc != 0:
        raise IndexError(f"orc_delta_stepping_timed rc={rc}")
    return dist[:n], secs.value


def delta_stepping(off, tgt, w, start_node: int, delta: float, native=False):
    n = off.size - 1
    dist = np.empty(max(n, 1), np.float32)
    wv = np.ascontiguousarray(w, np.float32)
    rc = (native_lib() if native else lib()).orc_delta_stepping(n, off, _tgt(tgt), wv if wv.size else np.zeros(1, np.float32), start_node,
                                  delta, dist)
    if rc != 0:
        raise IndexError(f"orc_delta_stepping rc={rc}")
    return dist[:n]


def sssp_fixed_point(off, tgt, w, start_node: int):
    """the least fixed point of d[v] = min(d[u] + w) in f32 (Dijkstra): the intended result of delta-stepping"""
    n = off.size - 1
    dist = np.empty(max(n, 1), np.float32)
    wv = np.ascontiguousarray(w, np.float32)
    L = lib()
    L.orc_sssp_fixed_point.argtypes = [C.c_uint32, _u32p, _u32p, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"),
                                       C.c_uint64, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]
    L.orc_sssp_fixed_point.restype = C.c_int
    rc = L.orc_sssp_fixed_point(n, off, _tgt(tgt), wv if wv.size else np.zeros(1, np.float32), start_node, dist)
    if rc != 0:
        raise IndexError(f"orc_sssp_fixed_point rc={rc}")
    return dist[:n]


def stale_check_misfires(dist, delta: float):
    """Reached nodes whose final distance d satisfies d < delta * (usize)(d / delta) in f32: the reference files
    them in a bin it then refuses to process (sssp.rs:126 vs :192), so its result is not the fixed point there."""
    d = np.asarray(dist, np.float32)
    fin = d < np.float32(3.0e38)
    dl = np.float32(delta)
    b = (d[fin] / dl).astype(np.int64)
    bad = d[fin] < (dl * b.astype(np.float32)).astype(np.float32)
    out = np.zeros(d.size, bool)
    out[np.flatnonzero(fin)[bad]] = True
    return out


def triangle_count(off, tgt, threads: int = 1, native=False) -> int:
    return int((native_lib() if native else lib()).orc_triangle_count(off.size - 1, off, _tgt(tgt), threads))


def greedy_degree_partition(off, concurrency: int):
    out = np.empty(2 * concurrency, np.uint32)
    k = lib().orc_greedy_degree_partition(off.size - 1, off, concurrency, out)
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(k)]
