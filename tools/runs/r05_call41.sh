#!/bin/bash
# round 5, GPU call 41: the graph_mate front's load_micros / __repr__ (new) with the rest of its acceptance checks
export TMPDIR=/tmp
timeout 30 python -m pytest tests/test_gpu_graph_mate.py -x -q -m gpu 2>&1 | tail -3
