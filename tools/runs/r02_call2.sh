#!/bin/bash
# round 2, GPU call 2: first run of the hub-row order emulation: unit tests, parity at 22/24/26, bench timing
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_order.py -m gpu -x -q -s > $OUT/pytest_hub.log 2>&1; tail -15 $OUT/pytest_hub.log
for s in 22 24 26; do
  timeout 600 python tools/parity_pagerank.py --scale $s --mode pb > $OUT/parity${s}_pb.json 2> $OUT/parity$s.err
  python - <<PY
import json
d=json.load(open("$OUT/parity${s}_pb.json"))
print("scale $s max_rel", d["max_rel_vs_reference"], "rows>1e-5", d["rows_over_1e-5"], "iters", d["device"]["iterations"], d["device"]["next_call_s"])
for c in d["by_in_degree"]: print("  ", c["in_degree"], c["rows"], c["max_rel"], c["rms_rel"])
PY
done
timeout 600 python bench.py --cpu-sweeps 0 > $OUT/bench26.json 2> $OUT/bench26.err; cat $OUT/bench26.json
GM_PB_HUB_DEG=0 timeout 600 python bench.py --cpu-sweeps 0 > $OUT/bench26_nohub.json 2> $OUT/bench26_nohub.err; cat $OUT/bench26_nohub.json
timeout 300 python bench.py --cpu-sweeps 0 --scale 22 > $OUT/bench22.json 2>&1; tail -1 $OUT/bench22.json
GM_PB_HUB_DEG=0 timeout 300 python bench.py --cpu-sweeps 0 --scale 22 > $OUT/bench22_nohub.json 2>&1; tail -1 $OUT/bench22_nohub.json
