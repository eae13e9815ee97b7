#!/bin/bash
# round 2, GPU call 62: SSSP with the 0.75 m .. 3 m band at scale 22 / 24 / 26, tests
OUT=gpurun_out/r02bi; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "sssp or delta" > $OUT/pytest.log 2>&1; grep -a "passed\|failed" $OUT/pytest.log | tail -2
for sc in 22 24 26; do
for band in new old; do
if [ $band = old ]; then A="GM_SSSP_ADAPT=$(python -c "print(round((1<<$sc)*16/5/1e6,3), round((1<<$sc)*16*0.75/1e6,3), sep=',')")"; else A="X=1"; fi
env $A timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 4 --sssp-scale $sc > $OUT/s${sc}_$band.json 2>/dev/null
echo "scale $sc $band: $(python -c "import json; d=json.load(open('$OUT/s${sc}_$band.json'))['sssp']; print(round(d['ms'],2))") ms"
done; done
