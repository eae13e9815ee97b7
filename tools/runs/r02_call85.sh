#!/bin/bash
# round 2, GPU call 85: pseudo-random gaps between the bins' areas of the value stream: correctness, then sweep times
# in alternating fresh processes (plain / gaps / contiguous pages + gaps)
OUT=gpurun_out/r02cf; mkdir -p $OUT; export TMPDIR=/tmp
GM_PB_BIN_GAP=4096 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hub_order.py -m gpu -x -q -k "page_rank or pb or hub or long_chains" > $OUT/pytest.log 2>&1; grep -a "passed\|failed" $OUT/pytest.log | tail -1
for k in 1 2 3; do
for cfg in "X=1" "GM_PB_BIN_GAP=4096" "GM_PB_BIN_GAP=4096 GM_PB_VALS_CONTIG=1"; do
env $cfg timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] run $k', d['ms_per_step'], d['roofline']['frac'])"
done; done
