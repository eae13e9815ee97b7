#!/bin/bash
# round 5, GPU call 21: which hub rows of the Unsorted scale-26 graph differ from the sequential sums after one sweep?
timeout 900 python tools/debug_unsorted.py 26 2>&1 | tail -6 | cut -c1-600
timeout 900 python tools/debug_unsorted.py 24 2>&1 | tail -5 | cut -c1-400
