#!/bin/bash
# round 5, GPU call 8: partitioned sweeps with the hub kernels BESIDE part 0's accumulate kernel (every part joins them)
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q -m gpu -k "multi or partition or piece or part" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
for r in 0 1 4; do timeout 600 python bench.py --emulate-parts 8 --emulate-rank $r --cpu-sweeps 0 --algos 0 2> $OUT/emu8_$r.err | tail -1 > $OUT/emu8_$r.json; python -c "import json; d = json.loads(open('$OUT/emu8_$r.json').read()); print('emulated rank $r of 8:', d['ms_per_step'])" || tail -5 $OUT/emu8_$r.err; done
for r in 0 1; do timeout 600 python bench.py --emulate-parts 2 --emulate-rank $r --cpu-sweeps 0 --algos 0 2> $OUT/emu2_$r.err | tail -1 > $OUT/emu2_$r.json; python -c "import json; d = json.loads(open('$OUT/emu2_$r.json').read()); print('emulated rank $r of 2:', d['ms_per_step'])" || tail -5 $OUT/emu2_$r.err; done
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t8 -o t -- python bench.py --emulate-parts 8 --emulate-rank 1 --cpu-sweeps 0 --algos 0 > $OUT/t8.log 2>&1
python tools/timeline.py $OUT/t8 1 | cut -c1-120
timeout 600 python bench.py --gpus 2 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --algos 0 2> $OUT/gloo2.err | tail -1 > $OUT/gloo2.json; python -c "import json; d = json.loads(open('$OUT/gloo2.json').read()); print('2 gloo ranks on one GPU:', d['ms_per_step'], d['config']['final_sweep_error'])" || tail -5 $OUT/gloo2.err
timeout 600 python bench.py --gpus 4 --backend gloo --single-device 1 --scale 22 --cpu-sweeps 0 --algos 0 2> $OUT/gloo4.err | tail -1 > $OUT/gloo4.json; python -c "import json; d = json.loads(open('$OUT/gloo4.json').read()); print('4 gloo ranks on one GPU (one stream per part):', d['ms_per_step'], d['config']['final_sweep_error'])" || tail -5 $OUT/gloo4.err
timeout 300 python bench.py --scale 22 --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale 22, one rank:', d['ms_per_step'], d['config']['final_sweep_error'])"
find $OUT -name "*.db" -size +8M -delete
