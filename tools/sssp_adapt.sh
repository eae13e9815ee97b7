#!/bin/bash
# usage: tools/sssp_adapt.sh "<width> <lo,hi>" ... : SSSP scale-24 time and work for adaptive threshold steps
for spec in "$@"; do
  set -- $spec
  w=$1; a=$2
  st=$(GM_SSSP_WIDTH=$w GM_SSSP_ADAPT=$a GM_SSSP_STATS=1 timeout -s KILL 300 python tools/bench_algos.py --skip pr,wcc,tc 2>&1 >/dev/null | grep "sssp:" | tail -1)
  ms=$(GM_SSSP_WIDTH=$w GM_SSSP_ADAPT=$a timeout -s KILL 300 python tools/bench_algos.py --skip pr,wcc,tc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read())['sssp']; print(round(d['ms'],2), d['fixed_point_le'], d['fixed_point_attained'], d['reached'])")
  echo "width=$w adapt=$a -> $ms | $st"
done
