#!/bin/bash
# round 5, GPU call 26: triangle count at scale 24: the six tc_rows_kernel launches one by one, and the item size
OUT=gpurun_out/r05t; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace -d $OUT/t -o t -- python tools/bench_algos.py --profile 1 --skip prapi,wcc,sssp > $OUT/rec.json 2> $OUT/err.txt
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob("gpurun_out/r05t/t/**/*.db", recursive=True)[0])
rows = [(n, (e - s) / 1000.0, gx, ld) for n, s, e, gx, ld in c.execute("select name, start, end, grid_x, lds_size from kernels order by start") if "tc_rows_kernel" in n or "tc_count_kernel" in n]
for n, d, gx, ld in rows[-7:]:
    print(f"{n.split('(')[0][-34:]:36s} {d:9.1f} us  grid {gx:8d}  lds {ld}")
PY
timeout 600 python tools/tc_ab.py 24 "" "GM_TC_ITEM=1024" "GM_TC_ITEM=4096" "GM_TC_ITEM=512" "GM_TC_K=262144" 2>&1 | grep -a "best of"
find $OUT -name "*.db" -delete
