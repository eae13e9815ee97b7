#!/bin/bash
# round 3, GPU call 58: register groups of the accumulate kernel's value stream in flight (2 = the old pipeline's bytes in flight, 4, 6): A/B in one process
timeout 300 python tools/ab_depth.py 26 2>&1 | grep -a "^round"
timeout 300 python tools/ab_depth.py 22 2>&1 | grep -a "^round"
