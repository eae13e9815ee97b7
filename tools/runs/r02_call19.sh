#!/bin/bash
# round 2, GPU call 19: instruction counters of the PB kernels (is pb_hub_kernel issue-bound?)
OUT=gpurun_out/r02t; mkdir -p $OUT; export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | tr ' ' '_')
  GM_PB_HUB_FORK=0 timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$n -o pmc -- python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 > $OUT/pmc_$n.log 2>&1
done
python tools/pmc_collect.py $OUT/pmc_insts.json $OUT/pmc_*/ > /dev/null
python - <<PY
import json
d=json.load(open("$OUT/pmc_insts.json"))
for k,v in d.items():
    if 'pb_' in k and ('hub' in k or 'accum' in k or 'bin_kernel' in k):
        print(k[:40], {a: round(b) for a,b in v.items()})
PY
find $OUT -name "*.db" -size +5M -delete
