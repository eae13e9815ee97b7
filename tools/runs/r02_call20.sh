#!/bin/bash
# round 2, GPU call 20: SSSP round kernel with a lane-per-word front end
OUT=gpurun_out/r02u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
timeout 600 python tools/bench_algos.py --skip prapi,wcc,tc > $OUT/sssp.json 2> $OUT/sssp.err; python -c "
import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('sssp ms', d['ms'], 'relax/s', d['relaxed_edges_per_s'], d['parity'], d['roofline']['frac'])"
timeout 300 python tools/stress_sssp.py 22 5 2>&1 | tail -2
bash tools/runs/r02_call16.sh > /dev/null 2>&1
python - <<PY
import re
rows=[]
for l in open('gpurun_out/r02p/sssp_dispatches.txt'):
    m=re.match(r'.*start\s+([\d.]+) us\s+dur\s+([\d.]+) us', l)
    if m: rows.append((float(m.group(1)), float(m.group(2))))
body=rows[2:]
rk=[body[i][1] for i in range(0,len(body),3)]; ck=[body[i][1] for i in range(1,len(body),3)]
print('round', len(rk), round(sum(rk)), 'chunk', round(sum(ck)), 'span', round(rows[-1][0]+rows[-1][1]))
print('round', [round(x) for x in rk])
PY
