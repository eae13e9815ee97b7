#!/bin/bash
# round 6, GPU call 16: differential fuzz of the final library (a7520f35, 3c8d66f2, c03c6d43, c94a0cfe) against the oracle and the SSSP schedule stress, as at the end of round 5
OUT=gpurun_out/r06o; mkdir -p $OUT; export TMPDIR=/tmp
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
( time timeout 300 python tools/fuzz_parity.py 200 601 3000 20000 ) > $OUT/fuzz_a.log 2>&1; tail -4 $OUT/fuzz_a.log
( time GM_PB_HOT=0 timeout 300 python tools/fuzz_parity.py 100 602 3000 20000 ) > $OUT/fuzz_b.log 2>&1; tail -4 $OUT/fuzz_b.log
( time GM_TC_K=50 GM_SSSP_COOP=4 GM_SSSP_CHUNK=64 timeout 300 python tools/fuzz_parity.py 200 603 3000 40000 ) > $OUT/fuzz_c.log 2>&1; tail -4 $OUT/fuzz_c.log
( time timeout 200 python tools/fuzz_parity.py 25 604 30000 300000 ) > $OUT/fuzz_d.log 2>&1; tail -4 $OUT/fuzz_d.log
( time timeout 300 python tools/stress_sssp.py 22 8 ) > $OUT/stress.log 2>&1; tail -4 $OUT/stress.log
