#!/bin/bash
# round 2, GPU call 69: dispatch lists of the final SSSP and triangle-count kernels
OUT=gpurun_out/r02bp; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -f csv -- python tools/bench_algos.py --profile 1 --skip prapi,wcc > $OUT/kt.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for tag, out in (('sssp_', 'sssp_dispatches.txt'), ('tc_', 'tc_dispatches.txt')):
    sel = [r for r in rows if tag in r['Kernel_Name']]
    t0 = int(sel[0]['Start_Timestamp'])
    tot = {}
    with open('$OUT/' + out, 'w') as o:
        for r in sel:
            k = re.search(tag + r'\w+', r['Kernel_Name']).group(0)
            s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += d
            o.write(f"{k:28s} start {s:10.1f} us  dur {d:9.1f} us  grid {r.get('Grid_Size_X', '?'):>10s}\n")
    print(tag, {k: (c, round(t)) for k, (c, t) in tot.items()}, 'span us', round((int(sel[-1]['End_Timestamp']) - t0) / 1e3))
PY
rm -rf $OUT/kt
