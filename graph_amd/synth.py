"""Synthetic R-MAT / Graph500 inputs generated and turned into CSR on the device.

The reference has no generator (its benches download LDBC Graph500 files,
crates/builder/benches/common/mod.rs:15-41); SURVEY §8d fixes the recipe: A=.57 B=.19 C=.19
D=.05, edge factor 16, seed 42, scrambled ids, directed, not deduplicated, CsrLayout::Sorted.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib, vp
from .prelude import CsrLayout, DeviceCsr, Direction


def rmat_edges(scale: int, seed: int = 42, edge_factor: int = 16, device: int = 0):
    m = edge_factor << scale
    dev = torch.device("cuda", device)
    src = torch.empty(m, dtype=torch.int32, device=dev)
    dst = torch.empty(m, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().gm_rmat_edges_device(scale, seed, 0, m, src.data_ptr(), dst.data_ptr(), device,
                                         vp(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()
    return src, dst


def rmat_edge_range(scale: int, seed: int, first: int, count: int, device: int = 0):
    """edges [first, first + count) of the same R-MAT edge list (the generator is counter-based: any range, on any device)"""
    dev = torch.device("cuda", device)
    src = torch.empty(count, dtype=torch.int32, device=dev)
    dst = torch.empty(count, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib().gm_rmat_edges_device(scale, seed, first, count, src.data_ptr(), dst.data_ptr(), device,
                                         vp(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()
    return src, dst


def rmat_weights(m: int, seed: int = 44, device: int = 0):
    dev = torch.device("cuda", device)
    w = torch.empty(m, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().gm_rmat_weights_device(seed, 0, m, w.data_ptr(), device,
                                           vp(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()
    return w


def build_csr(n: int, src: torch.Tensor, dst: torch.Tensor, direction: Direction, layout: CsrLayout,
              weights: torch.Tensor | None = None, device: int = 0) -> DeviceCsr:
    h = vp()
    torch.cuda.synchronize(device)
    check(lib().gm_csr_build_device(n, src.numel(), src.data_ptr(), dst.data_ptr(),
                                    weights.data_ptr() if weights is not None else 0, int(direction), int(layout),
                                    device, C.byref(h)))
    return DeviceCsr(h)
