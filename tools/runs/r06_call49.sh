#!/bin/bash
# round 6, GPU call 49: the Python front's slices with and without source flags against the single engine (new test)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "python_front" 2>&1 | tail -12 | cut -c1-220
