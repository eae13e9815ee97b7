#!/bin/bash
# round 2, GPU call 36: kernel breakdown of the triangle count
OUT=gpurun_out/r02aj; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -f csv -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,sssp > $OUT/kt.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    m = re.search(r'(tc_\w+|relabel\w*|\w*scan\w*|\w*sort\w*)', k)
    name = m.group(1) if m else k[:40]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot.setdefault(name, [0, 0.0]); tot[name][0] += 1; tot[name][1] += d
for k, (c, t) in sorted(tot.items(), key=lambda x: -x[1][1])[:14]:
    print(f'{k:40s} {c:4d} {t:12.1f} us')
PY
rm -rf $OUT/kt
