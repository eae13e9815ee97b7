#!/bin/bash
# round 4, GPU call 13: LDS-only barriers in the hub kernels (the prefetched loads stay in flight): single-row probe, hub tests, traces
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
for kind in long seq; do
  timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/$kind -o t -- python tools/hub_probe.py $kind 9000 16384 131072 1048576 > $OUT/$kind.log 2>&1
  python - <<PY
import sqlite3, glob
c = sqlite3.connect(glob.glob("$OUT/$kind/**/*.db", recursive=True)[0])
cur = [(n.split("(")[0].split("::")[-1], (e - s) / 1000.0) for n, s, e in c.execute("select name, start, end from kernels order by start") if "pb_hub" in n]
print("$kind:", " ".join(f"{d:.1f}" for n, d in cur), "us")
PY
done
for sc in 26 22; do for fork in 1 0; do
  GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${sc}_$fork -o trace -- python bench.py --cpu-sweeps 0 --scale $sc > $OUT/trace${sc}_$fork.log 2>&1
  echo "== scale $sc fork $fork: $(grep -a '^{' $OUT/trace${sc}_$fork.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")"
  DB=$(find $OUT/trace${sc}_$fork -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 8 | cut -c1-150 | grep "gm::pb_[abh]"
done; done
find $OUT -name "*.db" -size +20M -delete
