#!/bin/bash
# usage: tools/exp_env.sh "<ENV=.. ENV=..>|<scale> <steps> <warmup> [bench args]" ...
for spec in "$@"; do
  envs="${spec%%|*}"; rest="${spec#*|}"
  set -- $rest
  scale=$1; steps=$2; warm=$3; shift 3
  out=$(env $envs timeout -s KILL 300 python bench.py --scale $scale --steps $steps --warmup $warm --cpu-sweeps 0 "$@" 2>&1 | tail -1)
  echo "$out" | python -c "
import sys, json
line = sys.stdin.read().strip()
try:
    d = json.loads(line); r = d['roofline']
    print('[$envs] scale=$scale [$*] engine=%s ms/step=%.4f kern_ms=%.4f frac=%.4f GTEPS=%.1f' % (d['config']['engine'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['value']))
except Exception as e:
    print('[$envs] scale=$scale FAILED:', line[-300:])
"
done
