#!/bin/bash
# round 6, GPU call 39: rows below the hub threshold whose terms are all EQUAL (leaf fans) against the reference's left-to-right sum
export TMPDIR=/tmp
timeout 600 python tools/leaf_fan_probe.py 18 2>&1 | tail -14
GM_PB_HUB_DEG=256 timeout 600 python tools/leaf_fan_probe.py 18 2>&1 | tail -14
