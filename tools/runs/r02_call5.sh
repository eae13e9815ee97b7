#!/bin/bash
# round 2, GPU call 4: hub groups as extra bins + pb_hub_kernel on a second stream: correctness + timing + kernel stats
OUT=gpurun_out/r02e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -m gpu -x -q -s > $OUT/pytest_hub.log 2>&1; tail -6 $OUT/pytest_hub.log
for s in 26 24 22; do
  timeout 600 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench$s.json 2> $OUT/bench$s.err; python -c "
import json; d=json.loads(open('$OUT/bench$s.json').read().strip().splitlines()[-1]); print('scale $s hub   ', d['ms_per_step'], d['roofline']['frac'])"
  GM_PB_HUB_DEG=0 timeout 600 python bench.py --cpu-sweeps 0 --scale $s > $OUT/bench${s}_nohub.json 2> $OUT/bench${s}_nohub.err; python -c "
import json; d=json.loads(open('$OUT/bench${s}_nohub.json').read().strip().splitlines()[-1]); print('scale $s no hub', d['ms_per_step'], d['roofline']['frac'])"
done
for s in 26 22; do
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace$s -o trace -- python bench.py --cpu-sweeps 0 --scale $s > $OUT/trace$s.log 2>&1
DB=$(find $OUT/trace$s -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB 4 > $OUT/kernel_stats$s.txt
cat $OUT/kernel_stats$s.txt | cut -c1-50,105-160
done
timeout 600 python tools/parity_pagerank.py --scale 26 --mode pb > $OUT/parity26_pb.json 2> $OUT/parity26.err
python - <<PY
import json
d=json.load(open("$OUT/parity26_pb.json"))
print("scale 26 max_rel", d["max_rel_vs_reference"], "rows>1e-5", d["rows_over_1e-5"], "iters", d["device"]["iterations"], d["device"]["next_call_s"])
for c in d["by_in_degree"]: print("  ", c["in_degree"], c["rows"], c["max_rel"], c["rms_rel"])
PY
find $OUT -name "*.db" -size +20M -delete
