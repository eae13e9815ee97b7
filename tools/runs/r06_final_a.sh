#!/bin/bash
# round 6, final records (a): whole GPU suite, smoke(), kernel stats + PMC passes of the scale-26 sweep (stamped with this library's hash),
# two sweeps' timeline, the default bench line
OUT=gpurun_out/r06fa; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256; nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error" $OUT/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/profile.sh $OUT/prof --algos 0 > $OUT/profile.log 2>&1; head -8 $OUT/prof/kernel_stats.txt | cut -c1-150
python tools/timeline.py $OUT/prof/trace 2 > $OUT/timeline26.txt 2>&1; cat $OUT/timeline26.txt | cut -c1-100
python tools/pmc_traffic.py $OUT/prof/pmc_raw.json scale26_gpus1 9932111876 "round 6, tools/runs/r06_final_a.sh" > $OUT/pmc_traffic.txt 2>&1; tail -6 $OUT/pmc_traffic.txt; cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp $OUT/prof/pmc_raw.json $OUT/pmc_raw.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v.get('ms_result_left_on_device')) for k, v in d['extra'].items() if isinstance(v, dict)})"
find $OUT -name "*.db" -delete
