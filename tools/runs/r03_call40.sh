#!/bin/bash
# round 3, GPU call 40: whole GPU suite on the final binaries, two gloo ranks of bench.py sharing the GPU, the C-ABI multi entry with
# virtual ranks (GM_LOG)
OUT=gpurun_out/r03final; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 > $OUT/bench_scale22_two_ranks_gloo_one_gpu.json 2> $OUT/gloo2.err; tail -c 600 $OUT/bench_scale22_two_ranks_gloo_one_gpu.json; tail -3 $OUT/gloo2.err
timeout 600 python tools/multi_virtual.py 22 4 2> $OUT/multi_virtual_scale22_4ranks.txt >/dev/null; grep -a "==\|multi:" $OUT/multi_virtual_scale22_4ranks.txt | cut -c1-200
