#!/bin/bash
# round 5, GPU call 25: Unsorted scale 26, PB engine's hub rows against the REFORDER engine on the same inputs, every sweep of 130, 8 processes
for i in 1 2 3 4 5 6 7 8; do timeout 600 python tools/debug_unsorted.py 26 130 2>&1 | grep -a "differ\|done" | head -3 | cut -c1-300; done
