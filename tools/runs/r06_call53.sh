#!/bin/bash
# round 6, GPU call 53: sources with AT MOST ONE in-edge among the in-neighbours of RMAT rows (would a wider rule flag BASELINE rows?)
export TMPDIR=/tmp
LEAF_MAX_INDEG=1 timeout 900 python tools/leaf_sources_count.py 22 24 26 2>&1 | grep -a "scale\|in-degree"
LEAF_MAX_INDEG=2 timeout 900 python tools/leaf_sources_count.py 26 2>&1 | grep -a "scale\|in-degree"
