#!/bin/bash
# round 3, GPU call 49: one process, one plan, the value stream obtained in seven ways: is any of them fast where the arena's candidates are slow?
timeout 300 python tools/placement12.py 26 2>&1 | grep -a "bin kernel alone"
