#!/bin/bash
# round 4, GPU call 31: pb_hubseq_kernel with a counted memory pipeline (no vmcnt(0) in the loop)
OUT=gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"; }
for sc in 26 22; do
  for rep in 1 2; do timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc"; done
  for any in 0 1; do
    GM_PB_ANYORDER=$any timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace$sc$any -o trace -- python bench.py --cpu-sweeps 0 --algos 0 --scale $sc --steps 10 > $OUT/trace$sc$any.log 2>&1
    echo "timeline scale $sc, serial=$any"; python tools/timeline.py $OUT/trace$sc$any 1 > $OUT/timeline$sc$any.txt 2>&1; cat $OUT/timeline$sc$any.txt
  done
done
sed -e 's#gpurun_out/r04k#gpurun_out/r04z#' -e 's/for kind in seq long/for kind in seq/' tools/runs/r04_call11.sh > /tmp/probe.sh; bash /tmp/probe.sh
find $OUT -name "*.db" -delete
