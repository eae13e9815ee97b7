#!/bin/bash
# round 3, GPU call 42: arena / trim tests, the multi entry's residency for K = 1 after a K = 2 call
OUT=gpurun_out/r03zg; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_arena.py tests/test_gpu_multi.py -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|^E  " $OUT/pytest.log | tail -12
timeout 600 python tools/multi_virtual.py 24 4 2>&1 >/dev/null | grep -a "^==\|multi:" | cut -c1-170
