#!/bin/bash
# round 3, GPU call 18: debug of the equal-terms drift with the graded first block
OUT=gpurun_out/r03r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/hub_debug2.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/hub_debug2.txt
