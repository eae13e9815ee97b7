#!/bin/bash
# round 4, GPU call 8: pb_hublong_kernel : 32 terms per thread, one barrier per ordinary pass, adaptive long-row threshold; the hub kernels alone (GM_PB_HUB_FORK=0: in line) at scale 26 and 22
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
for sc in 26 22; do for fork in 1 0; do
  GM_PB_HUB_FORK=$fork timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace${sc}_$fork -o trace -- python bench.py --cpu-sweeps 0 --scale $sc > $OUT/trace${sc}_$fork.log 2>&1
  echo "== scale $sc fork $fork: $(grep -a '^{' $OUT/trace${sc}_$fork.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")"
  DB=$(find $OUT/trace${sc}_$fork -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB 8 | cut -c1-150 | grep "gm::pb_[abh]"
done; done
find $OUT -name "*.db" -size +20M -delete
