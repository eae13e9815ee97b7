#!/bin/bash
# round 2, GPU call 54: SSSP without an upper limit on the threshold step; the page_rank handle test
OUT=gpurun_out/r02ba; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sssp or delta or one_handle" --durations=5 > $OUT/pytest.log 2>&1; grep -a "passed\|failed\|long_path\|s call" $OUT/pytest.log | tail -8
GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 1 --reps 5 > $OUT/sssp24.json 2> $OUT/sssp24.err
python -c "import json; d=json.load(open('$OUT/sssp24.json'))['sssp']; print('scale 24', round(d['ms'],2), d['parity']['bit_exact_vs_oracle'])"; grep "sssp: setup" $OUT/sssp24.err | tail -1
timeout 300 python tools/stress_sssp.py 22 5 2>&1 | tail -1
