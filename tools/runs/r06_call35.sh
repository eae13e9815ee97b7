#!/bin/bash
# round 6, GPU call 35: how many of a plan's lane-walk terms are HOT records (GM_LOG=1 plan lines) — whole graph and a rank of 8
OUT=gpurun_out/r06ah; mkdir -p $OUT; export TMPDIR=/tmp
GM_LOG=1 timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --algos 0 --steps 5 --warmup 2 2>&1 | grep -a "hub groups\|hub rows\|hot source\|pb plan: seg\|value stream" | cut -c1-200
echo ---- rank 0 of 8
GM_LOG=1 timeout 300 python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank 0 --steps 5 --warmup 2 2>&1 | grep -a "hub groups\|hub rows\|hot source\|pb plan: seg\|value stream" | cut -c1-200
