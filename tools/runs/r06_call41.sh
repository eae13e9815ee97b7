#!/bin/bash
# round 6, GPU call 41: RMAT edges on node counts that are not powers of two — parity of the propagation-blocking engine on every row,
# default hub threshold and lower ones
export TMPDIR=/tmp
for hd in 4096 1024 512 256; do
GM_PB_HUB_DEG=$hd timeout 900 python tools/pad_n_probe.py 20 1048576 1100000 1234567 1500000 1777777 2000000 2>&1 | grep -a "^scale"
done
