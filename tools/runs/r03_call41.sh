#!/bin/bash
# round 3, GPU call 41: the C-ABI multi entry with 4 virtual ranks at scale 24 (slices large enough for propagation-blocking engines)
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/multi_virtual.py 24 4 2> $OUT/multi_virtual_scale24_4ranks.txt >/dev/null; grep -a "==\|multi:" $OUT/multi_virtual_scale24_4ranks.txt | cut -c1-200
