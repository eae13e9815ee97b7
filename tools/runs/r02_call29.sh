#!/bin/bash
# round 2, GPU call 29: SSSP tests (work split, handle reuse, bad weights), steady-state time, stress
OUT=gpurun_out/r02ad; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py tests/test_gpu_cpp_prelude.py -m gpu -x -q -k "sssp or delta or prelude" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 5 > $OUT/sssp24.json 2> $OUT/sssp24.err
python -c "import json; d=json.load(open('$OUT/sssp24.json'))['sssp']; print('scale 24', round(d['ms'],2), 'first', round(d['first_call_ms'],2))"; grep "sssp: setup" $OUT/sssp24.err | tail -2
GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 5 --sssp-scale 22 > $OUT/sssp22.json 2> $OUT/sssp22.err
python -c "import json; d=json.load(open('$OUT/sssp22.json'))['sssp']; print('scale 22', round(d['ms'],2), 'first', round(d['first_call_ms'],2))"; grep "sssp: setup" $OUT/sssp22.err | tail -1
GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 --sssp-scale 26 > $OUT/sssp26.json 2> $OUT/sssp26.err
python -c "import json; d=json.load(open('$OUT/sssp26.json'))['sssp']; print('scale 26', round(d['ms'],2), 'first', round(d['first_call_ms'],2))"; grep "sssp: setup" $OUT/sssp26.err | tail -1
timeout 300 python tools/stress_sssp.py 22 5 2>&1 | tail -2
