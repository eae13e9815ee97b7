#!/bin/bash
# round 6, GPU call 50 (last): the whole GPU suite + smoke on the tree (346 tests), the leaf-fan probe with the default plan (the rule on)
OUT=gpurun_out/r06an; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/leaf_fan_probe.py 18 2>&1 | grep -a "leaf fans\|fan of" | tee $OUT/leaf_default.txt
