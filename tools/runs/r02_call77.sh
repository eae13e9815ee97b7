#!/bin/bash
# round 2, GPU call 77: triangle count with the DAG kept in the handle
OUT=gpurun_out/r02bx; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "triangle or tc or graph_mate or prelude or robust" > $OUT/pytest_tc.log 2>&1; grep -a "passed\|failed" $OUT/pytest_tc.log | tail -2
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --oracle 1 --reps 5 > $OUT/tc24.json 2> $OUT/tc24.err
python -c "import json; d=json.load(open('$OUT/tc24.json'))['tc']; print('scale 24', round(d['ms'],2), 'first', round(d['first_call_ms'],2), d['parity']['bit_exact_vs_oracle'], d['roofline']['frac'])"
timeout -s KILL 600 python tools/bench_algos.py --skip prapi,wcc,sssp --tc-scale 22 --oracle 0 --reps 5 > $OUT/tc22.json 2> $OUT/tc22.err
python -c "import json; d=json.load(open('$OUT/tc22.json'))['tc']; print('scale 22', round(d['ms'],2), 'first', round(d['first_call_ms'],2))"
