#!/bin/bash
# round 3, GPU call 50: size of the arena's physical pieces (64 MiB / 256 MiB / 1 GiB) against the sweep, fresh processes interleaved on one box
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['config']['value_stream_placement']; print('$1', d['ms_per_step'], d['roofline']['frac'], 'draws', v['draws_timed'], 'best', v['draw_best_us'], 'worst', v['draw_worst_us'], 'grown', v['arena_grown_pieces'], 'plan', d['config']['plan_build_ms'], d['config']['plan_rebuild_ms'])"; }
for rep in 1 2 3; do for mib in 64 256 1024; do GM_ARENA_PIECE_MIB=$mib timeout 300 python bench.py --cpu-sweeps 0 2>/dev/null | tail -1 | line "pieces of $mib MiB:"; done; done
