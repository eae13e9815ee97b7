#!/bin/bash
# round 3, GPU call 19: the whole GPU suite (as the driver runs it), smoke, the default bench (cpu_baseline + parity measured in-run)
OUT=gpurun_out/r03s; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|^real|^E  " $OUT/pytest_gpu.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
( time timeout 600 python bench.py ) > $OUT/bench_default.log 2>&1; tail -4 $OUT/bench_default.log | cut -c1-3000
