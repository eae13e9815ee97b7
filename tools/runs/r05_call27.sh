#!/bin/bash
# round 5, GPU call 27: SSSP with the result left on the device (test + timing beside the host-result call)
OUT=gpurun_out/r05u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sssp" 2>&1 | tail -2
timeout 600 python tools/bench_algos.py --skip prapi,wcc,tc --oracle 2 > $OUT/sssp.json 2> $OUT/sssp.err; python -c "
import json; d=json.load(open('$OUT/sssp.json'))['sssp']; print('sssp scale 24: host result', round(d['ms'],3), 'best', round(d['best_ms'],3), '| device result', round(d['ms_result_left_on_device'],3), 'best', round(d['best_ms_result_left_on_device'],3), d['parity']['bit_exact_vs_oracle'])" || tail -5 $OUT/sssp.err
