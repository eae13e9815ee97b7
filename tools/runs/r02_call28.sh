#!/bin/bash
# round 2, GPU call 28: SSSP 64 sub-queues, short lists flattened over the wavefront
OUT=gpurun_out/r02ac; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_graph_mate.py -m gpu -x -q -k "sssp or delta" > $OUT/pytest_sssp.log 2>&1; tail -3 $OUT/pytest_sssp.log
run() { # name, env...
  name=$1; shift
  env "$@" GM_SSSP_TIMES=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 3 > $OUT/sssp_$name.json 2> $OUT/sssp_$name.err
  ms=$(python -c "import json; d=json.load(open('$OUT/sssp_$name.json'))['sssp']; print(round(d['ms'],2))")
  st=$(env "$@" GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 2>&1 >/dev/null | grep "sssp:" | tail -1)
  echo "$name: $ms ms | $(grep 'sssp: setup' $OUT/sssp_$name.err | tail -1) | $st"
}
run default X=1
run fixed1 GM_SSSP_WIDTH=1 GM_SSSP_ADAPT=0,0
run fixed05 GM_SSSP_WIDTH=0.5 GM_SSSP_ADAPT=0,0
run adapt_100_400 GM_SSSP_ADAPT=100,400
run adapt_200_800 GM_SSSP_ADAPT=200,800
run adapt_25_100 GM_SSSP_ADAPT=25,100
run w16_adapt_100_400 GM_SSSP_WIDTH=0.0625 GM_SSSP_ADAPT=100,400
run coop16 GM_SSSP_COOP=16
timeout 600 python tools/bench_algos.py --skip prapi,wcc,tc > $OUT/sssp.json 2> $OUT/sssp.err; python -c "
import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('sssp ms', d['ms'], 'relax/s', d['relaxed_edges_per_s'], d['parity'], d['roofline']['frac'])"
timeout 300 python tools/stress_sssp.py 22 5 2>&1 | tail -2
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,tc > $OUT/kt.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'sssp' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
tot = {}
with open('$OUT/sssp_dispatches.txt', 'w') as o:
    for r in rows:
        k = re.search(r'sssp_\w+', r['Kernel_Name']).group(0)
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += d
        o.write(f"{k:28s} start {(int(r['Start_Timestamp'])-t0)/1e3:10.1f} us  dur {d:9.1f} us\n")
print({k: (c, round(t)) for k, (c, t) in tot.items()}, 'span us', round((int(rows[-1]['End_Timestamp']) - t0) / 1e3))
PY
find $OUT -name "*.db" -delete; rm -rf $OUT/kt
