#!/bin/bash
export TMPDIR=/tmp
GM_ARENA_SITES=1 timeout 300 python tools/csr_debug.py 21 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
GM_ARENA=0 timeout 300 python tools/csr_debug.py 21 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
