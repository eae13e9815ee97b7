#!/bin/bash
# round 2, GPU call 11: the whole GPU suite + smoke
OUT=gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_all.log 2>&1; tail -30 $OUT/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
