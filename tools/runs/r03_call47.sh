#!/bin/bash
# round 3, GPU call 47: six private plans one after the other in one process (released / kept alive): does a rebuild change the level?
for keep in 0 1; do echo "== earlier plans kept alive: $keep"; timeout 300 python tools/placement11.py 26 6 $keep 2>&1 | grep -a "^plan"; done
