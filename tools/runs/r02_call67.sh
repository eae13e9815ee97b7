#!/bin/bash
# round 2, GPU call 67 (scale 22): timeline of the kernels of a few PageRank sweeps (gaps between bin / accumulate / hub)
OUT=gpurun_out/r02bn; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/kt -o kt -f csv -- python bench.py --cpu-sweeps 0 --steps 6 --warmup 3 ${BENCH_ARGS} > $OUT/kt.log 2>&1
grep -a '^{' $OUT/kt.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
python - <<PY
import csv, glob, re
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pb_' in r['Kernel_Name'] and 'keys' not in r['Kernel_Name'] and 'plan' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = {'pb_bin_kernel':'bin','pb_accum_kernel':'accum','pb_hub_kernel':'hub','pb_hot_gather_kernel':'hot','pb_err_kernel':'err'}
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in names)]
sel = sel[-5*len(names)*1:]  # last few sweeps
t0 = int(sel[0]['Start_Timestamp'])
with open('$OUT/pb_timeline.txt','w') as o:
    for r in sel[-25:]:
        k = [v for kk,v in names.items() if kk in r['Kernel_Name']][0]
        s = (int(r['Start_Timestamp'])-t0)/1e3; e = (int(r['End_Timestamp'])-t0)/1e3
        line = f"{k:6s} start {s:10.1f} end {e:10.1f} dur {e-s:8.1f} us  queue {r.get('Queue_Id','?')}"
        o.write(line+'\n'); print(line)
PY
rm -rf $OUT/kt
