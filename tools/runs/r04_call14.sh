#!/bin/bash
# round 4, GPU call 14: the default bench line with `extra` (WCC / SSSP / TC in the same run), PMC traffic stamped with the library's hash
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-6000; tail -3 $OUT/bench.err | cut -c1-300
bash tools/profile.sh $OUT/prof > $OUT/profile.log 2>&1; tail -14 $OUT/profile.log | cut -c1-160
python tools/pmc_traffic.py $OUT/prof/pmc_raw.json scale26_gpus1 9932111876 "round 4, tools/runs/r04_call14.sh" > $OUT/pmc_traffic.txt 2>&1; tail -5 $OUT/pmc_traffic.txt; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
python tools/timeline.py $OUT/prof/trace 1 > $OUT/timeline.txt 2>&1; head -20 $OUT/timeline.txt
timeout 300 python bench.py --cpu-sweeps 0 --algos 0 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('traffic on the line:', d['roofline']['traffic'], d['roofline']['traffic_source'])"
