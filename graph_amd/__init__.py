"""graph_amd — MI355X-native hot path of neo4j-labs/graph's `crates/algos` behind its prelude API.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/graph_mi355x.h), prelude.py (host
mirror of graph::prelude), engine.py / distributed.py (resident PageRank engine and the 1-D
vertex-range multi-GPU driver over torch.distributed/RCCL), synth.py (device R-MAT inputs).
"""
from ._lib import GraphMI355XError, LIB_PATH, build, device_count, lib  # noqa: F401
