#!/bin/bash
# round 6, GPU call 25: smaller lane-walk groups in a slice (about one per CU) — hub tests, then A/B on emulated ranks of 8 / 4 / 2
# (GM_PB_HUB_SLICE_GROUPS=0: full-size groups, 224 default, 448) and the timeline of rank 0 of 8
OUT=gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q -m gpu -x 2>&1 | tail -2
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); h = d['config'].get('hub_rows_in_reference_order') or {}; print('$1', d['ms_per_step'], d['config']['value_stream_placement'].get('level'), 'groups', h.get('hub_groups'), 'long', h.get('long_rows'))"; }
for rank in 0 6; do for gsz in 0 224 448 0 224; do GM_PB_HUB_SLICE_GROUPS=$gsz timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $rank 2>> $OUT/bench.err | tail -1 | line "rank $rank of 8 slice_groups=$gsz"; done; done
for gsz in 0 224; do GM_PB_HUB_SLICE_GROUPS=$gsz timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 4 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 4 slice_groups=$gsz"; done
for gsz in 0 224; do GM_PB_HUB_SLICE_GROUPS=$gsz timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 2 --emulate-rank 0 2>> $OUT/bench.err | tail -1 | line "rank 0 of 2 slice_groups=$gsz"; done
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t0 -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 > $OUT/t0.log 2>&1
python tools/timeline.py $OUT/t0 2 > $OUT/timeline_rank0.txt 2>&1; head -16 $OUT/timeline_rank0.txt | cut -c1-110
find $OUT -name "*.db" -delete
