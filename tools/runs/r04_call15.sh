#!/bin/bash
# round 4, GPU call 15: the whole GPU suite (arena size classes + growth test, exact hub kernels), the scale-22 / 26 sweep timelines
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -s > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3; grep -a "address space handed out\|scale 26\|scale 24," $OUT/pytest.txt | cut -c1-250
for sc in 22 26; do
  timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/trace$sc -o trace -- python bench.py --cpu-sweeps 0 --scale $sc --steps 10 > $OUT/trace$sc.log 2>&1
  python tools/timeline.py $OUT/trace$sc 1 > $OUT/timeline$sc.txt 2>&1; cat $OUT/timeline$sc.txt
done
find $OUT -name "*.db" -delete
