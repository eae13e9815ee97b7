#!/bin/bash
# round 2, GPU call 40: HBM traffic and instruction counters of tc_rows_kernel / tc_count_kernel
OUT=gpurun_out/r02am; mkdir -p $OUT; export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  GM_TC_K=524288 timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o pmc -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,sssp > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import glob, sqlite3
res={}
for db in glob.glob("$OUT/pmc_*/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    try:
        q="select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
        for kernel, counter, total, cnt in c.execute(q):
            if 'tc_' in kernel:
                import re
                k=re.search(r'tc_\w+', kernel).group(0)
                res.setdefault(k, {})[counter]=(total, cnt)
    except Exception as e:
        print('ERR', db, e, tabs[:8])
for k,v in res.items():
    print(k)
    for a,(t,c) in sorted(v.items()): print('   ', a, f'{t:.5g}', 'dispatches', c)
PY
find $OUT -name "*.db" -delete
