#!/bin/bash
# round 6, GPU call 43: sources without in-edges among the in-neighbours of RMAT rows (would a "many constant sources" rule flag BASELINE rows?)
export TMPDIR=/tmp
timeout 900 python tools/leaf_sources_count.py 22 24 26 2>&1 | grep -a "scale\|in-degree"
