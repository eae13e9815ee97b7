#!/bin/bash
# round 3, GPU call 52: 32 MiB against 64 MiB arena pieces, fresh processes alternating on one box, twelve each
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['config']['value_stream_placement']; print('$1', d['ms_per_step'], d['roofline']['frac'], 'draws', v['draws_timed'], 'best', v['draw_best_us'], 'grown', v['arena_grown_pieces'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do for mib in 32 64; do GM_ARENA_PIECE_MIB=$mib timeout 300 python bench.py --cpu-sweeps 0 --steps 10 --warmup 3 2>/dev/null | tail -1 | line "pieces of $mib MiB:"; done; done
