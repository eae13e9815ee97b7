#!/bin/bash
# round 5, GPU call 6: the whole GPU suite + smoke on the library with the multi-workgroup long rows and the drawn TC items
OUT=gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
