#!/bin/bash
# round 6, GPU call 55: the engine choice below 2^24 edges looks at the rule for rows of equal terms too — the new test, the hub-order
# file, first-call times of the drop-in call at scale 22 / 24 (the check walks lists once per handle)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -q -m gpu -x -s -k "small_graph" 2>&1 | grep -a "default call\|rule off\|passed\|failed\|rror" | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_hub_order.py tests/test_gpu_parity.py tests/test_gpu_multi.py -q -m gpu -x 2>&1 | grep -a "passed\|failed\|rror" | tail -3
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --oracle 0 2>/dev/null | python -c "import sys, json; print(json.load(sys.stdin)['page_rank_api'])" | cut -c1-300
timeout 600 python tools/bench_algos.py --skip wcc,sssp,tc --prapi-scale 24 --oracle 0 2>/dev/null | python -c "import sys, json; print(json.load(sys.stdin)['page_rank_api'])" | cut -c1-300
