#!/bin/bash
# round 3, GPU call 60: arena address ranges reused best-fit with split remainders — REVERTED: the bench process of this call did not finish within 300 s (the tests and six plan rebuilds before it did); exact-size reuse stays
timeout 1200 python -m pytest tests/test_gpu_arena.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -q 2>&1 | tail -1
timeout 300 python tools/placement11.py 26 6 1 2>&1 | grep -a "^plan"
timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'])"
