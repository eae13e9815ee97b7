"""ctypes binding of libgraph_mi355x.so (the C ABI of include/graph_mi355x.h).

The shared library is built in-tree by ``graph_amd.build()`` / ``__graft_entry__.build()``
(hipcc, gfx950).  There is no CPU fallback: if the library is missing, or a call fails because no
MI355X is visible, this module raises — it never routes work anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRAPH_MI355X_LIB: another build of the same ABI (tools/ablate.py loads the measurement library `make measure` produces)
LIB_PATH = os.environ.get("GRAPH_MI355X_LIB") or os.path.join(_HERE, "libgraph_mi355x.so")
CSRC = os.path.join(_HERE, "csrc")

GM_OK = 0
GM_ERR_INVALID, GM_ERR_RANGE, GM_ERR_HIP, GM_ERR_NOMEM, GM_ERR_UNSUPPORTED = -1, -2, -3, -4, -5

u64, u32, i32, f32, f64, vp = C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_double, C.c_void_p
PP = C.POINTER(vp)

# name -> (restype, argtypes); must list every function declared in include/graph_mi355x.h
SIGNATURES = {
    "gm_last_error": (C.c_char_p, []),
    "gm_abi_version": (i32, []),
    "gm_device_count": (i32, [C.POINTER(i32)]),
    "gm_csr_upload_u32": (i32, [vp, vp, vp, u64, u64, i32, PP]),
    "gm_csr_upload_u64": (i32, [vp, vp, vp, u64, u64, i32, PP]),
    "gm_csr_wrap_device": (i32, [u64, u64, u64, u64, u64, i32, PP]),
    "gm_csr_free": (None, [vp]),
    "gm_csr_trim": (i32, [vp]),
    "gm_csr_set_source_flags": (i32, [vp, u64, u64]),
    "gm_trim": (i32, [i32]),
    "gm_arena_info": (i32, [i32, vp]),
    "gm_arena_va_info": (i32, [i32, vp]),
    "gm_csr_node_count": (u64, [vp]),
    "gm_csr_edge_count": (u64, [vp]),
    "gm_csr_device": (i32, [vp]),
    "gm_csr_offsets_ptr": (u64, [vp]),
    "gm_csr_targets_ptr": (u64, [vp]),
    "gm_csr_weights_ptr": (u64, [vp]),
    "gm_csr_download": (i32, [vp, vp, vp, vp]),
    "gm_csr_degrees": (i32, [vp, vp]),
    "gm_csr_build_device": (i32, [u64, u64, u64, u64, u64, i32, i32, i32, PP]),
    "gm_csr_build_host": (i32, [u64, u64, vp, vp, vp, i32, i32, i32, PP]),
    "gm_csr_slice_rows": (i32, [vp, u64, u64, vp, u32, u32, PP]),
    "gm_csr_slice_rows_map": (i32, [vp, u64, u64, u64, PP]),
    "gm_csr_to_undirected": (i32, [vp, i32, PP]),
    "gm_csr_relabel_by_degree": (i32, [vp, PP, vp]),
    "gm_page_rank": (i32, [vp, vp, u64, f64, f32, i32, vp, C.POINTER(u64), C.POINTER(f64)]),
    "gm_page_rank_directed": (i32, [vp, vp, u64, f64, f32, i32, vp, C.POINTER(u64), C.POINTER(f64)]),
    "gm_pr_create": (i32, [vp, u64, u64, u64, f32, PP]),
    "gm_pr_create_with": (i32, [vp, u64, u64, u64, u64, f32, i32, PP]),
    "gm_pr_engine": (i32, [vp]),
    "gm_pr_destroy": (None, [vp]),
    "gm_pr_init": (i32, [vp, u64, u64, vp]),
    "gm_pr_sweep": (i32, [vp, u64, u64, u64, u64, vp]),
    "gm_pr_sweep_tiles": (i32, [vp, u64, u64, u64, vp]),
    "gm_pr_sweep_fixup": (i32, [vp, u64, u64, u64, vp]),
    "gm_pr_algorithmic_bytes": (u64, [vp]),
    "gm_page_rank_multi": (i32, [vp, vp, vp, u32, u64, f64, f32, vp, C.POINTER(u64), C.POINTER(f64)]),
    "gm_page_rank_multi_slices": (i32, [vp, vp, vp, u64, vp, u32, u64, f64, f32, vp, C.POINTER(u64), C.POINTER(f64)]),
    "gm_pr_tile_count": (u64, [vp]),
    "gm_pr_plan_info": (i32, [vp, vp, u32]),
    "gm_pr_part_geometry": (i32, [vp, vp, vp]),
    "gm_pr_set_parts": (i32, [vp, vp, u64]),
    "gm_pr_sweep_bin": (i32, [vp, u64, u64, u64, vp]),
    "gm_pr_set_bin_regions": (i32, [vp, vp, vp, vp, u64, u32]),
    "gm_pr_sweep_bin_region": (i32, [vp, u64, u32, vp]),
    "gm_pr_sweep_hot": (i32, [vp, u64, vp]),
    "gm_pr_sweep_accum": (i32, [vp, u64, u64, u64, u64, i32, vp]),
    "gm_wcc_afforest": (i32, [vp, vp, u64, u64, vp]),
    "gm_wcc_baseline": (i32, [vp, vp]),
    "gm_wcc_init_labels": (i32, [u64, u64, i32, vp]),
    "gm_wcc_link_rows": (i32, [vp, vp, u64, u64, u64, vp]),
    "gm_sssp_delta_stepping": (i32, [vp, u64, f32, vp]),
    "gm_sssp_init_distances": (i32, [u64, u64, u64, i32, vp]),
    "gm_sssp_relax_rows": (i32, [vp, u64, u64, u64, u64, vp]),
    "gm_triangle_count": (i32, [vp, C.POINTER(u64)]),
    "gm_rmat_edges_device": (i32, [u32, u64, u64, u64, u64, u64, i32, vp]),
    "gm_rmat_weights_device": (i32, [u64, u64, u64, u64, i32, vp]),
}


class GraphMI355XError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libgraph_mi355x status {status}: {message}")
        self.status = status


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into graph_amd/libgraph_mi355x.so (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  Our library
    needs the same SONAME; whichever copy is loaded first wins for us, but a later `import torch`
    would still map its bundled copy by path -> two HIP runtimes in one process (streams and
    events of one are meaningless to the other, device discovery can fail).  So when torch is
    installed but not imported yet, map its bundled runtime first; then both sides share it."""
    import sys

    if "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # no torch: the system ROCm runtime is used


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). graph_amd has no CPU fallback."
            )
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != GM_OK:
        msg = lib().gm_last_error()
        raise GraphMI355XError(status, msg.decode() if msg else "")


def device_count() -> int:
    c = i32(0)
    check(lib().gm_device_count(C.byref(c)))
    return c.value
