#!/bin/bash
# round 4, GPU call 41: counters of the SSSP kernels at scale 24, one counter per pass (the heavy round's chunk kernel: what is it made of?)
OUT=gpurun_out/r04zj; mkdir -p $OUT; export TMPDIR=/tmp
for c in TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_BUSY_CYCLES FETCH_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $OUT/p_$c -o p -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,tc > $OUT/p_$c.log 2>&1 || echo "$c: pass failed"
done
python - <<PY
import sqlite3, glob, os
for d in sorted(glob.glob("$OUT/p_*")):
    if not os.path.isdir(d): continue
    dbs = glob.glob(d + "/**/*.db", recursive=True)
    if not dbs: print(os.path.basename(d), "no db"); continue
    c = sqlite3.connect(dbs[0])
    try:
        rows = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
    except Exception as e:
        print(os.path.basename(d), "no counters", e); continue
    ch = sorted((v for k, n, v in rows if "sssp_chunk" in k), reverse=True)
    rd = sorted((v for k, n, v in rows if "sssp_round" in k), reverse=True)
    name = rows[0][1] if rows else "?"
    print(f"{name}: chunk top3 {[f'{x:.3g}' for x in ch[:3]]} sum {sum(ch):.3g} | round top3 {[f'{x:.3g}' for x in rd[:3]]} sum {sum(rd):.3g}")
PY
find $OUT -name "*.db" -delete
