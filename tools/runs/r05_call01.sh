#!/bin/bash
# round 5, GPU call 1 — evidence on round 4's final library, before anything changes:
#   per-CALL kernel stats + FETCH / WRITE / L2 hit-miss counters of WCC scale 22, SSSP scale 24 (calls 1, 2, 3), TC scale 24;
#   SQ / LDS counters of the PageRank sweep's kernels (pb_accum_kernel above all);
#   the hub threshold swept (4096 / 2048 / 1024) at scale 26: time and distance from the reference;
#   PageRankConfig::default() on the device against the reference at scale 22.
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/host.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host.txt 2>/dev/null
sha256sum graph_amd/libgraph_mi355x.so > $OUT/lib.sha256
rocprofv3 -L > $OUT/counters_list.txt 2>&1
have() { for c in "$@"; do grep -qw "$c" $OUT/counters_list.txt && echo -n "$c "; done; }
# --- the other three algorithms: one traced run, then counter passes (counters in their own runs, --kernel-trace only)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $OUT/algos_trace -o t -- python tools/bench_algos.py --profile 1 > $OUT/algos_record.json 2> $OUT/algos_trace.err
tail -c 600 $OUT/algos_record.json; echo
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
  cs=$(have $set); [ -z "$cs" ] && { echo "no counter of [$set] on this box"; continue; }
  tag=$(echo $cs | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 500 rocprofv3 --pmc $cs --kernel-trace -d $OUT/algos_pmc_$tag -o p -- python tools/bench_algos.py --profile 1 > $OUT/algos_pmc_$tag.json 2> $OUT/algos_pmc_$tag.err
  echo "pass [$cs]: rc $?"
done
ALGOS_PROFILE_JSON=$OUT/algos_profile.json python tools/algos_profile.py $OUT/algos_record.json $OUT/algos_trace $OUT/algos_pmc_* > $OUT/algos_profile.txt 2>&1
grep -a "^## \|total " $OUT/algos_profile.txt | cut -c1-220 | head -60
# --- the sweep's kernels: SQ / LDS / TCP counters (rocprofv3 --pmc serialises the dispatches: every kernel is measured ALONE)
B="python bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sweeps 0 --algos 0"
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_WAVES SQ_LEVEL_WAVES"; do
  cs=$(have $set); [ -z "$cs" ] && { echo "no counter of set $k"; k=$((k+1)); continue; }
  timeout -s KILL 300 rocprofv3 --pmc $cs --kernel-trace -d $OUT/sweep_pmc_$k -o p -- $B > $OUT/sweep_pmc_$k.log 2>&1
  echo "sweep pass $k [$cs]: rc $?"; k=$((k+1))
done
python tools/pmc_collect.py $OUT/sweep_counters_raw.json $OUT/sweep_pmc_* > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a/sweep_counters_raw.json"))
for k, v in d.items():
    if any(s in k for s in ("pb_accum", "pb_bin_kernel", "pb_hubseq_kernel", "pb_hublong_kernel")):
        print(k[:40], {c: f"{x:.4g}" for c, x in v.items() if not c.endswith(("_total", "_dispatches"))})
PY
# --- where this box stands (untraced), then the hub threshold and the default-config gap
timeout 300 $B --steps 20 --warmup 5 --prewarm-ms 400 > $OUT/bench_plain.json 2> $OUT/bench_plain.err; tail -1 $OUT/bench_plain.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('plain:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'))"
timeout 600 python tools/hub_deg_sweep.py --scale 26 --degs 4096,2048,1024 > $OUT/hub_deg_sweep.json 2> $OUT/hub_deg_sweep.err; grep -a hub_deg $OUT/hub_deg_sweep.err | cut -c1-330
timeout 300 python tools/default_config_gap.py --scale 22 > $OUT/default_config_gap.json 2> $OUT/default_config_gap.err; cat $OUT/default_config_gap.json
find $OUT -name "*.db" -size +8M -delete
