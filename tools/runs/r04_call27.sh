#!/bin/bash
# round 4, GPU call 27: same-box A/B of the two round-4b changes (hub groups' hot terms off the stream; bin chunk 24576) at scales 22 / 24 / 26
export TMPDIR=/tmp
for sc in 24 22 26; do for rep in 1 2; do for cfg in "A=1" "GM_PB_HUB_HOT=0" "GM_PB_CHUNK=32768" "GM_PB_HUB_HOT=0 GM_PB_CHUNK=32768"; do
  env $cfg timeout 200 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc [$cfg]:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('draw_best_us'))"
done; done; done
