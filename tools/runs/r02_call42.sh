#!/bin/bash
# round 2, GPU call 42: the WCC / SSSP / TC records of the round — full-size oracle comparisons, rooflines, CPU legs,
# kernel stats, PMC traffic, the SSSP dispatch list
OUT=gpurun_out/r02ao; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/bench_algos.py --reps 5 > $OUT/algos.json 2> $OUT/algos.err; tail -c 600 $OUT/algos.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python tools/bench_algos.py --profile 1 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 40 > $OUT/algos_kernel_stats.txt
python - <<PY
import sqlite3, glob, re
db = glob.glob('$OUT/trace/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
src = 'kernels' if 'kernels' in tabs else None
rows = []
if src:
    cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
    name = 'name' if 'name' in cols else 'kernel_name'
    for n, s, e in c.execute(f"select {name}, start, end from {src} order by start"):
        if 'sssp' in n:
            rows.append((re.search(r'sssp_\w+', n).group(0), s, e))
if rows:
    t0 = rows[0][1]
    with open('$OUT/sssp_dispatches.txt', 'w') as o:
        for n, s, e in rows:
            o.write(f"{n:28s} start {(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us\n")
    print('sssp dispatches', len(rows), 'span us', round((rows[-1][2] - t0) / 1e3))
else:
    print('no kernels view', tabs[:12])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o pmc -- python tools/bench_algos.py --profile 1 > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_collect.py $OUT/algos_pmc_raw.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
grep -a "sssp_\|tc_\|wcc_" $OUT/algos_kernel_stats.txt | cut -c1-60,105-160
find $OUT -name "*.db" -size +20M -delete
