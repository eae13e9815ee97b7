#!/bin/bash
# round 5, GPU call 22: the Unsorted scale-26 discrepancy through the page_rank() API (the engine API matches REFORDER on every sweep)
show() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['max_rel_vs_reference'], d['rows_over_1e-5'], d['device']['iterations'], d['reference']['iterations'], d['worst_rows'][:2])"; }
timeout 900 python tools/parity_pagerank.py --scale 24 --layout unsorted --mode pb 2>/dev/null | show "scale 24 unsorted:"
timeout 900 python tools/parity_pagerank.py --scale 26 --layout unsorted --mode pb 2>/dev/null | show "scale 26 unsorted:"
GM_PB_LONG_PASSES=16 timeout 900 python tools/parity_pagerank.py --scale 26 --layout unsorted --mode pb 2>/dev/null | show "scale 26 unsorted, 16 passes per item:"
timeout 900 python tools/parity_pagerank.py --scale 26 --layout sorted --mode pb 2>/dev/null | show "scale 26 sorted:"
