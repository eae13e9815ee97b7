#!/bin/bash
for k in "$@"; do
  out=$(GM_TC_K=$k timeout -s KILL 300 python tools/bench_algos.py --skip prapi,wcc,sssp 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read())['tc']; print('K=$k', round(d['tc_ms'],1), 'ms', d['triangles'])"
done
