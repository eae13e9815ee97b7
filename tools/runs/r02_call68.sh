#!/bin/bash
# round 2, GPU call 68: scale-22 PageRank: rows per bin and hot-table size (the hot table is as large as a bin's slice of the value stream there)
OUT=gpurun_out/r02bo; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-sweeps 0 --scale 22 --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$name', d['ms_per_step'], 'frac', d['roofline']['frac'], 'hot', c['hot_sources'], 'entries', c['value_entries'], 'wgs', c['workgroups_per_sweep'])"
}
run default X=1
run rb12 GM_PB_RB=12
run rb13 GM_PB_RB=13
run hot8192 GM_PB_HOT=8192
run hot4096 GM_PB_HOT=4096
run rb12_hot8192 GM_PB_RB=12 GM_PB_HOT=8192
run rb12_hot32768 GM_PB_RB=12 GM_PB_HOT=32768
run rb13_hot8192 GM_PB_RB=13 GM_PB_HOT=8192
run nofork GM_PB_HUB_FORK=0
run default_again X=1
