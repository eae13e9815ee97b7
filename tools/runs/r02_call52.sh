#!/bin/bash
# round 2, GPU call 52: page_rank() with its buffers and engine parked in the handle
OUT=gpurun_out/r02ay; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc > $OUT/prapi22.json 2> $OUT/prapi22.err; cat $OUT/prapi22.json | cut -c1-420
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --prapi-scale 26 > $OUT/prapi26.json 2> $OUT/prapi26.err; cat $OUT/prapi26.json | cut -c1-420
