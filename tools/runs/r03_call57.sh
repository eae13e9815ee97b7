#!/bin/bash
# round 3, GPU call 57: 16384- against 32768-source tiles (write runs twice as long), private plans alternating in one process
timeout 300 python tools/placement13.py 26 3 2>&1 | grep -a "^tiles"
