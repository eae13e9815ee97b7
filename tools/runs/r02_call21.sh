#!/bin/bash
# round 2, GPU call 21: counters of the SSSP round kernel
OUT=gpurun_out/r02v; mkdir -p $OUT; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA_RDREQ_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o pmc -- python tools/bench_algos.py --skip prapi,wcc,tc --oracle 0 --reps 1 > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import glob, sqlite3, json
res={}
for db in glob.glob("$OUT/pmc_*/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    for kernel, counter, total, cnt in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
        if 'sssp_round' in kernel or 'sssp_chunk' in kernel:
            res.setdefault(kernel.split('(')[1][:30] if '(' in kernel else kernel[:30], {})[counter]=(total, cnt)
for k,v in res.items():
    print(k)
    for a,(t,c) in sorted(v.items()): print('   ', a, f'{t:.4g}', 'dispatches', c)
PY
find $OUT -name "*.db" -delete
