#!/bin/bash
# round 5, GPU call 5: pb_hublong_kernel, a long row over several workgroups: the adversarial / order
# tests (array_equal against the sequential sums), the single-row probe, an emulated rank of 8 that owns the 854,315-term row
OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error\|error" $OUT/pytest.txt | tail -5
for passes in 8 4 16; do
  GM_PB_LONG_PASSES=$passes timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/long$passes -o t -- python tools/hub_probe.py long 9000 16384 131072 1048576 > $OUT/long$passes.log 2>&1
  python - <<PY
import sqlite3, glob
dbs = glob.glob("$OUT/long$passes/**/*.db", recursive=True)
if dbs:
    c = sqlite3.connect(dbs[0])
    cur = [(n.split("(")[0].split("::")[-1], (e - s) / 1000.0) for n, s, e in c.execute("select name, start, end from kernels order by start") if "pb_hublong" in n]
    print("passes per item $passes:", " ".join(f"{d:.1f}" for n, d in cur), "us")
else:
    print("no trace for $passes"); print(open("$OUT/long$passes.log").read()[-600:])
PY
done
for r in 0 1 2; do timeout 600 python bench.py --emulate-parts 8 --emulate-rank $r --cpu-sweeps 0 --algos 0 2> $OUT/emu8_$r.err | tail -1 > $OUT/emu8_$r.json; python -c "import json; d = json.loads(open('$OUT/emu8_$r.json').read()); print('emulated rank $r of 8:', d['ms_per_step'], d['config']['hub_rows_in_reference_order'])" || tail -5 $OUT/emu8_$r.err; done
timeout 300 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale 26:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), d['config']['final_sweep_error'])"
find $OUT -name "*.db" -size +8M -delete
