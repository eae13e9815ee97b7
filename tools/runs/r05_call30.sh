#!/bin/bash
# round 5, GPU call 30: differential fuzz of the final library against the oracle (random awkward graphs, every algorithm, every
# layout) and the SSSP schedule stress, for the record (profiles/r05_fuzz_final.txt)
OUT=gpurun_out/r05y; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
( time timeout 300 python tools/fuzz_parity.py 300 501 3000 20000 ) > $OUT/fuzz_a.log 2>&1; tail -5 $OUT/fuzz_a.log
( time timeout 300 python tools/fuzz_parity.py 60 502 200000 3000000 ) > $OUT/fuzz_b.log 2>&1; tail -5 $OUT/fuzz_b.log
( time GM_TC_K=50 GM_SSSP_COOP=4 GM_SSSP_CHUNK=64 GM_PB_HUB_DEG=64 timeout 300 python tools/fuzz_parity.py 200 503 3000 40000 ) > $OUT/fuzz_c.log 2>&1; tail -5 $OUT/fuzz_c.log
( time timeout 300 python tools/stress_sssp.py 22 8 ) > $OUT/stress.log 2>&1; tail -4 $OUT/stress.log
