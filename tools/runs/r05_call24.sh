#!/bin/bash
# round 5, GPU call 24: does the hopping-boundary stress test catch the per-thread reads of the boundary sums?  (old = the library before the fix)
for i in 1 2 3; do GRAPH_MI355X_LIB=$PWD/build/libgraph_old.so timeout 600 python -m pytest tests/test_gpu_hub_adversarial.py -x -q -m gpu -k "hop_over" 2>&1 | grep -a "passed\|failed\|assert\|Error" | head -3 | cut -c1-250; done
echo "--- fixed library:"; for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_hub_adversarial.py -x -q -m gpu -k "hop_over" 2>&1 | tail -1; done
