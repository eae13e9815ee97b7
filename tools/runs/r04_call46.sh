#!/bin/bash
# round 4, GPU call 46: SSSP after the round's changes: every SSSP test, scales 20 / 22 / 24 / 26 against CSR order + whole heavy rounds
OUT=gpurun_out/r04zo; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -2 $OUT/pytest_sssp.log
for sc in 20 22 24 26; do for cfg in "GM_SSSP_ORDER=0" "X=1"; do
  env $cfg GM_SSSP_TIMES=1 timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 --sssp-scale $sc > $OUT/sssp.json 2> $OUT/sssp.err
  python -c "import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('scale $sc $cfg:', round(d['ms'],3), 'ms, first call', round(d['first_call_ms'],1), 'reached', d['reached'])"
done; done
timeout 600 python tools/stress_sssp.py 22 5 2>&1 | tail -2
