#!/bin/bash
# round 4, GPU call 37: segments of the value stream padded to 8 entries (32-byte sectors never shared by two workgroups'
# writes) against 4: bin kernel time, sweep time, bytes written
OUT=gpurun_out/r04zf; mkdir -p $OUT; export TMPDIR=/tmp
GM_PB_SEGPAD=8 timeout 600 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_parity.py -x -q -m gpu -k "not sssp and not wcc and not triangle" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'], d['config']['value_entries'])"; }
for sc in 26 22; do for rep in 1 2; do for pad in 4 8; do
  GM_PB_SEGPAD=$pad timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc pad $pad"
done; done; done
GM_PB_SEGPAD=8 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc -o pmc -- python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 --algos 0 > $OUT/pmc.log 2>&1
python tools/pmc_collect.py $OUT/pmc_raw.json $OUT/pmc > $OUT/collect.log 2>&1; python - <<PY
import json
d = json.load(open("$OUT/pmc_raw.json"))
for k, v in d.items():
    if "pb_bin" in k or "pb_accum" in k: print(k[:40], v)
PY
find $OUT -name "*.db" -delete
