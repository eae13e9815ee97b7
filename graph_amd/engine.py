"""Resident PageRank engine over torch device tensors (plumbing: memory + streams only).

Wraps gm_pr_create / gm_pr_init / gm_pr_sweep: the sweep kernels of graph_amd/csrc/pagerank.hip
run on torch's current HIP stream, so torch.cuda.Event brackets them exactly.
"""
from __future__ import annotations

import ctypes as C
import sys

import torch

from ._lib import check, lib, vp


def current_stream_ptr() -> vp:
    return vp(torch.cuda.current_stream().cuda_stream)


class PageRankEngine:
    """page_rank_iteration (crates/algos/src/page_rank.rs:113-168) over rows
    [row_begin, row_begin + n_local) of a graph with n_global nodes; arrays are torch tensors."""

    AUTO, PULL, PB, REFORDER = 0, 1, 2, 3

    def __init__(self, in_csr_handle, n_global: int, row_begin: int, out_degree_local: torch.Tensor,
                 damping_factor: float = 0.85, x_len: int | None = None, engine: int = 0):
        assert out_degree_local.dtype == torch.int32 and out_degree_local.is_cuda
        self._keep = (in_csr_handle, out_degree_local)
        h = vp()
        check(lib().gm_pr_create_with(in_csr_handle, n_global, row_begin, x_len if x_len is not None else n_global,
                                      out_degree_local.data_ptr(), damping_factor, engine, C.byref(h)))
        self._h = h
        self.n_local = int(out_degree_local.numel())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and sys is not None and not sys.is_finalizing():
            lib().gm_pr_destroy(h)

    @property
    def algorithmic_bytes(self) -> int:
        return int(lib().gm_pr_algorithmic_bytes(self._h))

    @property
    def engine(self) -> str:
        return {1: "pull", 2: "pb", 3: "reforder"}[int(lib().gm_pr_engine(self._h))]

    @property
    def tiles(self) -> int:
        return int(lib().gm_pr_tile_count(self._h))

    def plan_info(self) -> dict:
        """what the propagation-blocking plan costs and contains (gm_pr_plan_info); {} for the other engines"""
        if self.engine != "pb":
            return {}
        v = (C.c_uint64 * 23)()
        check(lib().gm_pr_plan_info(self._h, v, 23))
        keys = ("plan_bytes", "plan_build_us", "hub_rows", "hub_edges", "hub_in_degree", "hot_sources", "value_entries",
                "hot_edges", "scratch_bytes", "bins", "source_tiles", "segments", "hub_groups", "hot_tiers", "long_rows",
                "long_row_terms", "hub_seq_blocks", "draw_best_us", "draw_worst_us", "draws_timed", "arena_grown_pieces",
                "value_stream_from_arena", "hub_hot_edges")
        return dict(zip(keys, (int(x) for x in v)))

    def init(self, scores_local: torch.Tensor, x_local: torch.Tensor):
        check(lib().gm_pr_init(self._h, scores_local.data_ptr(), x_local.data_ptr(), current_stream_ptr()))

    def sweep(self, x_in: torch.Tensor, x_out_local: torch.Tensor, scores_local: torch.Tensor,
              err_out: torch.Tensor):
        """one synchronous sweep; err_out: f64[1] device tensor receiving this rank's L1 error share"""
        check(lib().gm_pr_sweep(self._h, x_in.data_ptr(), x_out_local.data_ptr(), scores_local.data_ptr(),
                                err_out.data_ptr(), current_stream_ptr()))

    def sweep_tiles(self, x_in, x_out_local, scores_local):
        check(lib().gm_pr_sweep_tiles(self._h, x_in.data_ptr(), x_out_local.data_ptr(), scores_local.data_ptr(),
                                      current_stream_ptr()))

    def sweep_fixup(self, x_out_local, scores_local, err_out):
        check(lib().gm_pr_sweep_fixup(self._h, x_out_local.data_ptr(), scores_local.data_ptr(), err_out.data_ptr(),
                                      current_stream_ptr()))

    # ---- a sweep in pieces (propagation-blocking engines): see include/graph_mi355x.h ------------------
    def part_geometry(self):
        """(rows per bin, source tile): row splits must be multiples of the first, x regions of the second"""
        r, t = C.c_uint64(0), C.c_uint64(0)
        check(lib().gm_pr_part_geometry(self._h, C.byref(r), C.byref(t)))
        return int(r.value), int(t.value)

    def set_parts(self, row_splits):
        arr = (C.c_uint64 * len(row_splits))(*[int(v) for v in row_splits])
        check(lib().gm_pr_set_parts(self._h, arr, len(row_splits) - 1))

    def sweep_bin(self, x_in: torch.Tensor, x_lo: int, x_hi: int):
        # propagate x_in[x_lo:x_hi] (whole source tiles) into the value stream
        check(lib().gm_pr_sweep_bin(self._h, x_in.data_ptr(), x_lo, x_hi, current_stream_ptr()))

    def set_bin_regions(self, x_lo, x_hi, region, n_regions: int):
        """regions of x as LISTS of tile ranges (a rank-major exchanged vector: region k = group k of every rank)"""
        cnt = len(x_lo)
        lo = (C.c_uint64 * cnt)(*[int(v) for v in x_lo])
        hi = (C.c_uint64 * cnt)(*[int(v) for v in x_hi])
        rg = (C.c_uint32 * cnt)(*[int(v) for v in region])
        check(lib().gm_pr_set_bin_regions(self._h, lo, hi, rg, cnt, int(n_regions)))

    def sweep_bin_region(self, x_in: torch.Tensor, region: int):
        # propagate every tile range of one region into the value stream: ONE launch
        check(lib().gm_pr_sweep_bin_region(self._h, x_in.data_ptr(), int(region), current_stream_ptr()))

    def sweep_hot(self, x_in: torch.Tensor):
        check(lib().gm_pr_sweep_hot(self._h, x_in.data_ptr(), current_stream_ptr()))

    def sweep_accum(self, x_in: torch.Tensor, x_out_local: torch.Tensor, scores_local: torch.Tensor, part: int,
                    stage_hot: bool | None = None):
        """stage_hot None: on part 0 (the in-order, single-stream schedule)"""
        hot = (part == 0) if stage_hot is None else bool(stage_hot)
        check(lib().gm_pr_sweep_accum(self._h, x_in.data_ptr(), x_out_local.data_ptr(), scores_local.data_ptr(), part,
                                      1 if hot else 0, current_stream_ptr()))
