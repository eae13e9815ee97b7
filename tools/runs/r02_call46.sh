#!/bin/bash
# round 2, GPU call 46: per-launch times of the triangle count (row ranges, preparation)
OUT=gpurun_out/r02as; mkdir -p $OUT; export TMPDIR=/tmp
GM_TC_SHAPE=512,8,4 timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -f csv -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,sssp > $OUT/kt.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    k = r['Kernel_Name']
    if 'tc_' in k or 'scan' in k.lower():
        m = re.search(r'(tc_\w+)', k)
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print(f"{(m.group(1) if m else k[:30]):28s} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>10s} lds {r.get('LDS_Block_Size', r.get('LDS_Block_Size_v','?')):>8s} {d:10.1f} us")
PY
rm -rf $OUT/kt
