#!/bin/bash
# repeat the PB parity tests N times (after dirtying device memory with a large run) and show the first failure
N=${1:-10}
timeout -s KILL 120 python tools/ablate.py 24 3 > /dev/null 2>&1
for i in $(seq 1 $N); do
  out=$(timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -k "pb_engine_matches_exact or pb_split or partitioned_engines" 2>&1)
  if echo "$out" | grep -q "failed"; then
    echo "=== iteration $i FAILED"; echo "$out" | grep -vE "^\s*$" | head -80; exit 1
  fi
  echo "iteration $i ok: $(echo "$out" | tail -1)"
done
