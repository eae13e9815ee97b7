#!/bin/bash
# round 6, final records (d): after a change under graph_amd/csrc — the default line with both counter records in place (their hashes
# match the library), the emulated partition table, one emulated rank's timeline
OUT=gpurun_out/r06fd; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256; sha256sum bench.py >> $OUT/lib.sha256
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['value_stream_placement'].get('level'), d['config']['parity']['max_rel_vs_reference'], {k: (v.get('ms'), v.get('bit_exact'), v['roofline'].get('frac'), v['roofline'].get('traffic'), v.get('ms_result_left_on_device')) for k, v in d['extra'].items() if isinstance(v, dict)})"
timeout 2400 python tools/partition_emulated.py --scale 26 > $OUT/partition_emulated_scale26.json 2> $OUT/partition.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06fd/partition_emulated_scale26.json'))
for row in d['table']:
    print(row['gpus'], 'slowest', row['slowest_rank_ms'], 'fastest', row['fastest_rank_ms'], 'exchange model', row['exchange_ms_model'], 'projected speed-up', row['projected_speedup'])
PY
for rank in 0 6; do
timeout -s KILL 300 rocprofv3 --kernel-trace -d $OUT/t$rank -o t -- python bench.py --scale 26 --cpu-sweeps 0 --emulate-parts 8 --emulate-rank $rank --no-piece-events > $OUT/t$rank.log 2>&1
python tools/timeline.py $OUT/t$rank 2 > $OUT/timeline_rank$rank.txt 2>&1; head -16 $OUT/timeline_rank$rank.txt | cut -c1-100
done
find $OUT -name "*.db" -delete
