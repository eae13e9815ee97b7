#!/bin/bash
# round 2, GPU call 16: per-dispatch times of one SSSP run at scale 24 (where do the 24 ms of kernels go?)
OUT=gpurun_out/r02p; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o sssp -- python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 > $OUT/trace.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1); echo $F
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
ks=[r for r in rows if 'sssp' in r['Kernel_Name']]
ks.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(ks[0]['Start_Timestamp'])
out=[]
for r in ks:
    name=r['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::','')
    out.append((name[:22], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
open("$OUT/sssp_dispatches.txt","w").write("\n".join(f"{n:24s} start {s:10.1f} us  dur {d:9.1f} us" for n,s,d in out))
print(len(out), 'dispatches; span', out[-1][1]+out[-1][2], 'us')
rk=[d for n,s,d in out if 'round' in n]; ck=[d for n,s,d in out if 'chunk' in n]; ak=[d for n,s,d in out if 'advance' in n]
print('round', len(rk), sum(rk), 'chunk', len(ck), sum(ck), 'advance', len(ak), sum(ak))
print('round durs', [round(x) for x in rk])
print('chunk durs', [round(x) for x in ck])
PY
rm -rf $OUT/trace
