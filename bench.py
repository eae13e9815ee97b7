#!/usr/bin/env python3
"""bench.py — PageRank pull sweeps on synthetic R-MAT, the metric of BASELINE.json.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL.  Started under torch.distributed.run (WORLD_SIZE set) it is one
rank of the job; started plainly (`python bench.py --gpus 8`) it launches its own N ranks through
torch.distributed.run on 127.0.0.1 and relays rank 0's JSON line.

A *step* is one PageRank sweep (page_rank_iteration, crates/algos/src/page_rank.rs:113-168) over the
whole graph: K timed sweeps after W warm-up sweeps, bracketed by barrier + synchronize, MAX over
ranks.  value = m * K / seconds in GTEPS (whole job).  Inputs are resident in HBM when the timed
region starts (R-MAT generated and turned into a Sorted in-CSR on the device).

N = 1: the whole graph on one GPU.  N > 1: 1-D vertex-range partition (reference's greedy in-degree
partitioner), replicated out_scores, one RCCL all-gather per sweep — strong scaling (fixed graph).

Extra objects in the JSON line:
  roofline      the sweep's kernels (propagation blocking: pb_bin_kernel, then pb_accum_kernel with pb_hubseq_kernel /
                pb_hublong_kernel beside it; below 2^24 edges pr_tile_kernel): algorithmic bytes per launch (8m + 20n + 4
                over the rank's rows) / their average duration measured here with HIP events on the launch stream;
                `traffic` = HBM bytes per sweep from profiles/pmc_traffic.json IF that record was measured on the very
                library this process loaded (sha256 stamp), else null
  cpu_baseline  the oracle's restatement of the reference's threaded path (oracle/graph_oracle.c:
                orc_page_rank_chunked; kind "port") timed on this box's host cores on the same graph
  extra         N = 1: the reference app's other three algorithms in the same run (crates/app/src/app.rs:124-153 times all four):
                wcc_afforest RMAT scale-22, delta_stepping and global_triangle_count scale-24 — device time, byte-model
                roofline, threaded CPU leg, bit-exactness (tools/bench_algos.py; --algos 0 skips it)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)   # PageRankConfig::DEFAULT_MAX_ITERATIONS
    ap.add_argument("--warmup", type=int, default=5)   # crates/app/src/app.rs:124-153: 5 warm-up runs
    ap.add_argument("--scale", type=int, default=26)   # BASELINE.json metric: RMAT scale-26
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--pretouch-frac", type=float, default=0.0, help="experiment: allocate and free this fraction of the free "
                    "device memory before anything else (does the placement level follow the driver's allocation history?)")
    ap.add_argument("--cpu-sweeps", type=int, default=20, help="sweeps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores (available_parallelism)")
    ap.add_argument("--parity", type=int, default=1, help="N = 1, inside the cpu_baseline leg: run the timed engine and the CPU "
                    "path to their fixed points and compare every row (about 20 s of host time at scale 26; 0 = skip)")
    ap.add_argument("--algos", type=int, default=1, help="N = 1, default scale only: WCC scale 22 / SSSP scale 24 / triangle count scale 24 "
                    "after the PageRank leg -> `extra` (about 15 s; 0 = skip)")
    ap.add_argument("--tc-oracle", type=int, default=1, help="`extra`: check the scale-24 triangle count against orc_triangle_count in "
                    "this process (about 24 s on 16 host cores; 0 = compare with the count tests/test_gpu_fullsize.py pins)")
    ap.add_argument("--relabel", type=int, default=0, help="(experimental) internal degree-ordered layout")
    ap.add_argument("--engine", choices=["auto", "pull", "pb"], default="auto")
    ap.add_argument("--prewarm-ms", type=float, default=400.0, help="untimed sweeps before the W warm-up steps so "
                    "the GPU leaves its idle clock state (sclk idles at 576 MHz; short runs otherwise vary by 10 %%)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                    "exercising the multi-rank path with several ranks on ONE device)")
    ap.add_argument("--single-device", type=int, default=0, help="debug: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--exchange-parts", type=int, default=0,
                    help="N > 1: regions of the out_scores exchange that overlap with the work (1 = one blocking all-gather "
                         "per sweep; 0 = automatic: 2)")
    ap.add_argument("--exchange", choices=["allgather", "sparse"], default="allgather",
                    help="N > 1: allgather = every rank receives every source's out_score (in regions overlapped with the "
                         "work); sparse = every pair of ranks exchanges only what the receiver's rows read (blocking; covered "
                         "by gloo tests with a stand-in engine, not yet run on RCCL)")
    ap.add_argument("--bin-pieces", type=int, default=-1,
                    help="overlapped exchange: 1 = propagate every region as it lands, 0 = one propagation launch per sweep "
                         "(only the accumulate is cut; measured no cheaper: P = 8 kernels 0.41 -> 0.48 / 0.46-0.49 ms), -1 = 1")
    ap.add_argument("--emulate-parts", type=int, default=0, help="debug (1 process): time only the row slice that "
                    "rank --emulate-rank of an N-way partition would own, without the exchange")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--no-piece-events", action="store_true", help="debug (with --emulate-parts): no second pass with event "
                    "pairs around the pieces of a sweep (roofline.achieved is then not measured: 0)")
    ap.add_argument("--launch-check", type=int, default=0, help="debug: every rank reports its rendezvous and exits "
                    "before touching a GPU (tests/test_bench_launch_cpu.py)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, the
    way the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...`).  Rank 0 prints the JSON line; it passes through unchanged."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def _device_note(torch, dev):
    """what the run landed on (sweep times differ by up to 20 % between boxes of the same model)"""
    try:
        p = torch.cuda.get_device_properties(dev)
        note = f"{p.name}, {p.multi_processor_count} CUs, {p.total_memory >> 30} GiB"
        for attr in ("clock_rate", "memory_clock_rate"):
            if hasattr(p, attr):
                note += f", {attr} {getattr(p, attr) // 1000} MHz"
        return note
    except Exception as exc:  # informational only
        return f"unknown ({exc})"


def _placement(plan, engine, scale):
    """How the value stream's memory was chosen (arena.hip / pb_scratch_create) and which LEVEL this process landed on: the
    bin kernel's rate by its byte model during the timed draws — at RMAT scale 26 the fast level is 4060-4500 GB/s, the slow
    one (about one process in four, box-dependent, profiles/r03_box_survey.txt) <= 3700."""
    out = {k: plan.get(k) for k in ("value_stream_from_arena", "draws_timed", "draw_best_us", "draw_worst_us", "arena_grown_pieces")}
    us = plan.get("draw_best_us") or 0
    if us and plan.get("draws_timed", 0) > 0:
        moved = plan.get("value_entries", 0) * 6 + (1 << scale) * 4  # 2 B id + 4 B value per entry + the out_scores once
        gbs = moved / (us * 1e-6) / 1e9
        out["bin_kernel_GBps_by_byte_model"] = round(gbs, 0)
        out["level"] = "fast" if gbs >= 4000 else ("medium" if gbs >= 3700 else "slow")
        if out["level"] == "slow":
            out["note"] = "every candidate placement of the value stream stayed at the slow level in this process"
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_parts:
        raise SystemExit(self_launch(args))
    if args.launch_check:
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"launch_check": True, "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                              "master": os.environ.get("MASTER_ADDR"), "backend": args.backend}), flush=True)
        return
    import numpy as np
    import torch
    import torch.distributed as dist

    import graph_amd
    from graph_amd import synth
    from graph_amd.engine import PageRankEngine
    from graph_amd.prelude import CsrLayout, Direction
    from graph_amd._lib import check, lib, vp
    import ctypes as C

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available() or graph_amd.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    if args.single_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible GPUs, this node shows {torch.cuda.device_count()} "
                         f"(one process per GPU; --single-device 1 --backend gloo exercises the path on one)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.pretouch_frac > 0:
        free_b, _ = torch.cuda.mem_get_info(dev)
        t_pt = time.time()
        blocks = [torch.empty(int(free_b * args.pretouch_frac / 8), dtype=torch.uint8, device=dev) for _ in range(8)]
        del blocks
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        print(f"pretouch: {free_b * args.pretouch_frac / 2**30:.0f} GiB allocated and freed in {time.time() - t_pt:.2f} s", file=sys.stderr)
    scale, n = args.scale, 1 << args.scale
    m = args.edge_factor << scale
    mem_peak = [0]

    def mem_sample():
        """device bytes in use (everything on the device: torch, the library's arena, RCCL), sampled at the checkpoints below"""
        free_b, total_b = torch.cuda.mem_get_info(dev)
        mem_peak[0] = max(mem_peak[0], total_b - free_b)

    emu = args.emulate_parts if world == 1 else 0
    if emu:
        world, rank = emu, args.emulate_rank  # pretend; no process group exists
    if args.exchange_parts == 0:
        args.exchange_parts = 2
    if args.bin_pieces < 0:
        args.bin_pieces = 1
    sparse = world > 1 and args.exchange == "sparse" and not emu
    piecewise = world > 1 and args.exchange_parts > 1 and args.engine != "pull" and args.exchange != "sparse"
    ex = None
    t_build = time.time()
    if world == 1 or sparse or args.relabel:
        # N = 1: the whole graph on this GPU.  (Also the opt-in sparse exchange, whose layout is derived from the whole edge
        # list: gloo-tested with a stand-in engine, never the metric's path.)
        src, dst = synth.rmat_edges(scale, args.seed, args.edge_factor, local_rank)
        if args.relabel:
            # experiment: ids re-ranked by out-degree (descending) so the most-gathered out_scores are packed
            deg = torch.bincount(src, minlength=n)
            if args.relabel == 2:
                deg = deg + torch.bincount(dst, minlength=n)
            order = torch.argsort(deg, descending=True, stable=True)
            new_id = torch.empty(n, dtype=torch.int32, device=dev)
            new_id[order] = torch.arange(n, dtype=torch.int32, device=dev)
            src = new_id[src.long()]
            dst = new_id[dst.long()]
            del deg, order, new_id
        out_deg = torch.bincount(src, minlength=n).to(torch.int32)
        in_csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted, None, local_rank)
        edges = (src, dst) if sparse else None
        del src, dst
        torch.cuda.empty_cache()
        mem_sample()
        construction = "whole graph on the device"
        if world > 1:
            from graph_amd.distributed import greedy_degree_partition, pad_bounds

            off_host = np.empty(n + 1, np.uint32)
            check(lib().gm_csr_download(in_csr.handle, off_host.ctypes.data_as(vp), None, None))
            bounds, _ = pad_bounds(greedy_degree_partition(off_host, world), world, n)
            del off_host
    else:
        # N > 1 (and the one-device emulation of a rank): PARTITION-LOCAL construction — no rank generates the whole edge list
        # or builds the whole in-CSR (graph_amd/distributed.py:rank_local_rows): degree histograms of m / N edges summed over the
        # ranks, the reference's greedy in-degree ranges, then only the edges whose destination this rank owns
        from graph_amd.distributed import rank_local_rows

        in_csr, bounds, out_deg, edge_peak = rank_local_rows(scale, args.seed, rank, world, local_rank, args.edge_factor,
                                                             collective=not emu)
        edges = None
        mem_sample()
        construction = (f"partition-local: this rank generated and kept only the edges into its own rows "
                        f"({in_csr.m} of {m}); degree histograms of m / {world} edges per rank, summed with an all-reduce"
                        + (" (emulated rank: all of them scanned here)" if emu else ""))
    t_build = time.time() - t_build

    # ---- partition --------------------------------------------------------------------------
    if world == 1:
        local_csr, row_lo, n_local, stride = in_csr, 0, n, n
        out_deg_local = out_deg
        x_len = n
    else:
        from graph_amd.distributed import PiecewiseExchange, compact_exchange_layout, split_exchange_layout

        row_lo, row_hi = int(bounds[rank]), int(bounds[rank + 1])
        n_local = row_hi - row_lo
        # exchange only the out_scores of nodes that have out-edges (the others are never gathered)
        if piecewise:
            layout = split_exchange_layout(out_deg, bounds, parts=args.exchange_parts)
            node_map, x_len, stride = layout["node_map"], layout["x_len"], sum(layout["strides"])
        elif sparse:
            from graph_amd.distributed import SparseExchange, sparse_exchange_layout

            layout = sparse_exchange_layout(None, None, bounds, rank, edges=edges)
            edges = None
            node_map, x_len = layout["node_map"], layout["x_len"]
            stride = sum(c for q, c in enumerate(layout["recv_cnt"]) if q != rank)  # floats received per sweep
        else:
            node_map, send_counts, stride, send_rows_all = compact_exchange_layout(out_deg, bounds)
            send_rows = send_rows_all[rank]
            del send_rows_all
            x_len = world * stride
        h = vp()
        check(lib().gm_csr_slice_rows_map(in_csr.handle, row_lo, row_hi, node_map.data_ptr(), C.byref(h)))
        from graph_amd.prelude import DeviceCsr

        local_csr = DeviceCsr(h)
        # which slots of the exchanged vector hold nodes without in-edges (global knowledge: rank_local_rows' summed histogram), so that
        # the slice's plan flags the rows the whole graph's plan would (rows of many constant terms; RMAT: none)
        if getattr(in_csr, "no_in_edges", None) is not None:
            from graph_amd.distributed import source_flags

            local_csr.set_source_flags(source_flags(node_map, in_csr.no_in_edges, x_len))
        del node_map
        out_deg_local = out_deg[row_lo:row_hi].contiguous() if n_local else torch.zeros(1, dtype=torch.int32, device=dev)
        del in_csr
        in_csr = None
        torch.cuda.empty_cache()
        mem_sample()
    m_local = local_csr.m

    engine = PageRankEngine(local_csr.handle, n, row_lo, out_deg_local, 0.85, x_len=x_len,
                            engine={"auto": 2 if piecewise else 0, "pull": 1, "pb": 2}[args.engine])
    scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    mem_sample()
    if piecewise:
        gather = None
        if emu:  # stand-in for the collective: only this rank's slot of the region is refreshed
            def gather(dst_views, src, k):
                dst_views[rank].copy_(src)
        ex = PiecewiseExchange(engine, layout, rank, n_local, dev, gather=gather, split_bin=bool(args.bin_pieces))
        if emu:  # the slots of the ranks that do not exist: a typical out_score instead of zeros (rows that sum to 0 are not
            for buf in ex.x:  # what a sweep of the partitioned job walks)
                buf.fill_(1.0 / (16.0 * n))
        ex.start(scores)
    elif sparse and not emu:
        sx = SparseExchange(layout, n_local, dev)
        x, x_loc = sx.x, sx.x_loc

        def exchange(dst_buf):
            sx.exchange(0 if dst_buf is x[0] else 1)

        engine.init(scores, x_loc)
        exchange(x[0])
    else:
        x = [torch.zeros(x_len, dtype=torch.float32, device=dev) for _ in range(2)]
        x_loc = torch.zeros(max(n_local, 1), dtype=torch.float32, device=dev) if world > 1 else None
        x_send = torch.zeros(stride, dtype=torch.float32, device=dev) if world > 1 else None

        def exchange(dst_buf):
            x_send[: send_rows.numel()] = x_loc[send_rows]  # compaction gather (33 MB -> ~12 MB per rank at scale 26)
            if emu:  # stand-in: only this rank's slot is refreshed
                dst_buf[rank * stride:(rank + 1) * stride] = x_send
                return
            dist.all_gather_into_tensor(dst_buf, x_send)

        if world == 1:
            engine.init(scores, x[0])
        else:
            engine.init(scores, x_loc)
            exchange(x[0])
    cur = 0

    def step(timed_events=None):
        """timed_events: list that receives (start, end) event pairs around the sweep's kernels"""
        nonlocal cur
        if piecewise:
            ex.sweep(scores, err, timed_events)
            return
        out_local = x[1 - cur] if world == 1 else x_loc
        # one rank: the whole sweep through gm_pr_sweep, error included — on the propagation-blocking engine the error comes out of
        # the sweep's own launches (round 6: no pb_err_kernel; until then sweep_tiles + sweep_fixup, the events around the first)
        whole = world == 1
        if timed_events is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            if whole:
                engine.sweep(x[cur], out_local, scores, err)
            else:
                engine.sweep_tiles(x[cur], out_local, scores)
            pair[1].record()
            timed_events.append(pair)
        elif whole:
            engine.sweep(x[cur], out_local, scores, err)
        else:
            engine.sweep_tiles(x[cur], out_local, scores)
        if not whole:
            engine.sweep_fixup(out_local, scores, err)
        if world > 1:
            exchange(x[1 - cur])
        cur = 1 - cur

    def sync_all():
        if world > 1 and not emu:
            dist.barrier()
        torch.cuda.synchronize()

    if args.prewarm_ms > 0:  # clock ramp: not part of the W warm-up steps, nothing is measured here
        if world > 1 and not emu:
            # every rank must issue the same number of collectives: a fixed count, not a wall-clock loop
            for _ in range(64):
                step()
            torch.cuda.synchronize()
        else:
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    evs = []
    t0 = time.perf_counter()
    # A sweep in pieces (N > 1, emulated ranks) is timed WITHOUT event pairs around its pieces: every timed event record is a
    # barrier packet with a system-scope release, 6 of them per sweep cost an emulated rank of 8 25-45 us of its 0.55 ms
    # (tools/runs/r06_call28.sh) — the kernel time for `roofline` comes from a second pass below, outside the timed region.
    # One rank: one pair around the whole sweep, inside the timed region as the contract says.
    for k in range(args.steps):
        step(None if piecewise else evs)
    enqueue_seconds = time.perf_counter() - t0  # host time to enqueue the K steps (close to `seconds`: the host is the limit)
    sync_all()
    seconds = time.perf_counter() - t0
    event_steps = args.steps
    if piecewise and not (emu and args.no_piece_events):
        event_steps = min(args.steps, 64)  # (the same count on every rank: each step issues collectives)
        for k in range(event_steps):
            step(evs)
        sync_all()
    mem_sample()
    if piecewise:
        ex.finish()
    if world > 1 and not emu:
        tt = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        seconds = float(tt.item())
    mem_ranks = [mem_peak[0]]
    if world > 1 and not emu:
        mine = torch.tensor([mem_peak[0]], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        mem_ranks = [int(t.item()) for t in every]
    tile_ms_avg = sum(a.elapsed_time(b) for a, b in evs) / max(event_steps, 1)  # kernel time per sweep
    if world > 1 and not emu:
        dist.all_reduce(err, op=dist.ReduceOp.SUM)
    final_err = float(err.item())

    ms_per_step = seconds * 1e3 / max(args.steps, 1)
    gteps = m * args.steps / seconds / 1e9
    alg_bytes = engine.algorithmic_bytes  # 8*m_local + 20*n_local + 4
    achieved = alg_bytes / (tile_ms_avg * 1e-3) / 1e9 if tile_ms_avg > 0 else 0.0

    # PMC-derived HBM traffic per launch: only if the committed rocprofv3 --pmc summary was measured on the library this
    # process loaded (tools/pmc_traffic.py stamps the record with the sha256 of libgraph_mi355x.so)
    traffic, traffic_note = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            import hashlib

            rec = json.load(open(pmc_path)).get(f"scale{scale}_gpus{world}", {})
            with open(graph_amd.LIB_PATH, "rb") as fh:
                loaded = hashlib.sha256(fh.read()).hexdigest()
            if rec.get("hbm_bytes_per_launch") is None:
                traffic_note = "no PMC record for this configuration"
            elif rec.get("library_sha256") == loaded:
                traffic = rec["hbm_bytes_per_launch"]
                traffic_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this library ({loaded[:16]}), {rec.get('measured', 'undated')}"
            else:
                traffic_note = (f"profiles/pmc_traffic.json was measured on library {str(rec.get('library_sha256'))[:16]}, this process "
                                f"loaded {loaded[:16]}: not quoted")
        except Exception as exc:
            traffic, traffic_note = None, f"unreadable PMC record: {exc}"

    plan = engine.plan_info()  # propagation-blocking engines: what the resident plan cost and holds
    # The plan above was built first thing in this process (its time includes ~29 ms of one-time work of the HIP
    # runtime: loading the code object, first staging buffers).  What building it again costs: a private second plan
    # of the same CSR (GM_PB_NOCACHE), outside the timed region, dropped at once.
    plan_rebuild_ms = None
    if plan and world == 1 and not piecewise:
        os.environ["GM_PB_NOCACHE"] = "1"
        try:
            again = PageRankEngine(local_csr.handle, n, row_lo, out_deg_local, 0.85, x_len=x_len, engine=2)
            info = again.plan_info()
            plan_rebuild_ms = round(info["plan_build_us"] / 1e3, 2) if info else None
            again = None
        except Exception:
            plan_rebuild_ms = None
        finally:
            del os.environ["GM_PB_NOCACHE"]
    # parity of the timed engine is MEASURED below, in the cpu_baseline leg (the same CPU path run to its fixed point is
    # the checker of the engine that was just timed): null when that leg is skipped — nothing is quoted from a file
    parity = None
    result = {
        "metric": "pagerank_edges_per_sec",
        "value": round(gteps, 4),
        "unit": "GTEPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"PageRank pull sweep, RMAT scale-{scale} (A=.57 B=.19 C=.19 D=.05, edge factor "
                        f"{args.edge_factor}, seed {args.seed}), DirectedCsrGraph<u32> CsrLayout::Sorted, damping 0.85",
            "nodes": n, "edges": m, "step": "one sweep over all in-edges",
            "partition": "none" if world == 1 else
                         f"1-D vertex ranges (greedy in-degree), {world} ranks, all-gather of {stride * 4} B/rank/sweep "
                         f"(only nodes with out-edges)" + (f" in {args.exchange_parts} regions overlapped with the work"
                                                            + ", parts in order on one stream"
                                                            if piecewise else "") if not sparse else
                         f"1-D vertex ranges (greedy in-degree), {world} ranks, sparse pairwise exchange: this rank receives "
                         f"{stride * 4} B/sweep (the out_scores its rows read)",
            "construction": construction,
            # per rank (everything on its device: torch, the library's arena, RCCL), sampled after the build, the slice, the
            # plan and the timed sweeps; with --single-device every rank sees the same device
            "device_bytes_in_use_peak_per_rank": mem_ranks,
            "device": _device_note(torch, dev),
            "csr_build_s": round(t_build, 3), "final_sweep_error": final_err, "workgroups_per_sweep": engine.tiles, "engine": engine.engine,
            "plan_build_ms": round(plan["plan_build_us"] / 1e3, 2) if plan else None, "plan_rebuild_ms": plan_rebuild_ms,
            "plan_bytes": plan.get("plan_bytes") if plan else None, "scratch_bytes": plan.get("scratch_bytes") if plan else None,
            "hub_rows_in_reference_order": {k: plan[k] for k in ("hub_in_degree", "hub_rows", "hub_edges", "hub_groups", "long_rows",
                                                               "long_row_terms", "hub_seq_blocks", "hub_hot_edges")} if plan else None,
            "hot_sources": plan.get("hot_sources") if plan else None, "hot_tiers": plan.get("hot_tiers") if plan else None,
            "hot_edges": plan.get("hot_edges") if plan else None, "value_entries": plan.get("value_entries") if plan else None,
            "value_stream_placement": _placement(plan, engine, scale) if plan else None,
            "parity": parity,
        },
        "roofline": {
            "kernel": "pr_tile_kernel" if engine.engine == "pull" else "pb_bin_kernel+pb_accum_kernel+pb_hubseq_kernel+pb_hublong_kernel",
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(tile_ms_avg, 5),
            "edges_per_launch": m_local, "rows_per_launch": n_local,
        },
    }

    # ---- CPU baseline on rank 0 at N = 1 (bounded sample: a few sweeps of the same graph) -----
    if world == 1 and args.cpu_sweeps > 0:
        from oracle import oracle as O  # checker / timed CPU baseline only (see oracle/graph_oracle.c header)

        off_h = np.empty(n + 1, np.uint32)
        tgt_h = np.empty(m, np.uint32)
        check(lib().gm_csr_download(in_csr.handle, off_h.ctypes.data_as(vp), tgt_h.ctypes.data_as(vp), None))
        od_h = out_deg.cpu().numpy().astype(np.uint32)
        cores = args.cpu_threads or O.effective_cores()  # affinity mask capped by the cgroup CPU quota, as Rust's available_parallelism()
        cpu_s, _ = O.page_rank_chunked_timed(off_h, tgt_h, od_h, args.cpu_sweeps, 0.85, cores, spread=True)
        result["cpu_baseline"] = {
            "value": round(m * args.cpu_sweeps / cpu_s / 1e9, 4), "unit": "GTEPS", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_sweeps} sweeps of the same scale-{scale} graph (after 1 warm-up sweep), "
                      f"orc_page_rank_chunked_timed: 16384-node dynamic chunks, threads re-spawned per sweep, inputs "
                      f"first-touched by all threads (NUMA spread)",
            "ms_per_step": round(cpu_s * 1e3 / args.cpu_sweeps, 3),
            "build": O.timed_build_flags(),
        }
        if args.parity and not piecewise and not sparse:
            # The same CPU path run to its fixed point checks THE ENGINE THAT WAS TIMED (this process, this library, this
            # plan): PageRankConfig::new(200, 1e-10, 0.85) on both sides, every row compared (north_star: <= 1e-5 relative).
            engine.init(scores, x[0])
            sweeps_dev = 0
            for it in range(200):
                engine.sweep(x[it % 2], x[1 - it % 2], scores, err)
                sweeps_dev += 1
                if float(err.item()) < 1e-10:
                    break
            got = scores.cpu().numpy().astype(np.float64)
            t_ref = time.perf_counter()
            ref, it_ref, _ = O.page_rank_chunked(off_h, tgt_h, od_h, 200, 1e-10, 0.85, cores)
            t_ref = time.perf_counter() - t_ref
            ref = ref.astype(np.float64)
            rel = np.abs(got - ref) / ref
            deg_h = np.diff(off_h.astype(np.int64))
            hub = deg_h >= 4096
            result["config"]["parity"] = {
                "max_rel_vs_reference": float(rel.max()), "rows_over_1e-5": int((rel > 1e-5).sum()), "tolerance": 1e-5,
                "max_rel_rows_with_4096_or_more_in_edges": float(rel[hub].max()) if hub.any() else None,
                "max_in_degree": int(deg_h.max()), "device_sweeps": sweeps_dev, "reference_iterations": int(it_ref),
                "reference_seconds": round(t_ref, 2),
                "source": "measured in this run: the timed engine run on to its fixed point against oracle "
                          "orc_page_rank_chunked (page_rank.rs:113-168) on the host cores, PageRankConfig::new(200, 1e-10, 0.85)",
            }
    # ---- the reference app's other three algorithms, same run (N = 1, BASELINE's own scale only) -----------------------
    if world == 1 and not emu and args.algos and args.scale == 26 and args.cpu_sweeps > 0:
        engine = None
        ex = None
        del local_csr, scores, x
        in_csr = None
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_algos

        try:
            rec = bench_algos.measure(bench_algos.parse(["--oracle", "2", "--tc-oracle", str(args.tc_oracle), "--skip", "prapi", "--reps", "3"]))
            result["extra"] = {
                k: {"config": rec[k]["config"], "ms": round(rec[k]["ms"], 3), "best_ms": round(rec[k]["best_ms"], 3), "roofline": rec[k]["roofline"],
                    "bit_exact": rec[k]["parity"]["bit_exact_vs_oracle"] if rec[k]["parity"]["bit_exact_vs_oracle"] is not None
                    else rec[k]["parity"].get("equals_the_count_pinned_by_that_test"), "parity": rec[k]["parity"],
                    "cpu_baseline": rec[k].get("cpu_baseline"),
                    **({"triangles": rec[k]["triangles"]} if k == "tc" else {}),
                    **({"ms_result_left_on_device": round(rec[k]["ms_result_left_on_device"], 3)}
                       if k == "wcc" and rec[k].get("ms_result_left_on_device") else {}),
                    **({"relaxed_edges": rec[k]["relaxed_edges"], "first_call_ms": round(rec[k]["first_call_ms"], 3),
                        "ms_result_left_on_device": round(rec[k]["ms_result_left_on_device"], 3) if rec[k].get("ms_result_left_on_device") else None,
                        "second_call_ms_builds_the_ordered_lists": round(rec[k]["second_call_ms_builds_the_ordered_lists"], 3)}
                       if k == "sssp" else {})}
                for k in ("wcc", "sssp", "tc") if k in rec}
            result["extra"]["protocol"] = ("tools/bench_algos.py in this process after the PageRank leg: `ms` = MEAN of 3 calls through the "
                                           "prelude API (results downloaded) after warm-up calls, `best_ms` the fastest "
                                           "(crates/app/src/app.rs:124-153); SSSP: calls 3-5 on the "
                                           "handle (the first runs on the CSR's lists, the second builds the weight-ordered and "
                                           "transposed copies the later ones use: both timed beside `ms`)")
        except Exception as exc:  # the headline line must not depend on the extras
            result["extra"] = {"error": repr(exc)}
    if emu:
        result["config"]["emulated"] = f"rank {rank} of {world} on one device, exchange replaced by a local copy"
        result["config"]["host_enqueue_ms_per_step"] = round(enqueue_seconds * 1e3 / max(args.steps, 1), 5)
    if rank == 0 or emu:
        print(json.dumps(result), flush=True)
    if world > 1 and not emu:
        dist.barrier()
        dist.destroy_process_group()
    # release every device object explicitly before interpreter shutdown (a HIP call from a
    # destructor during Python finalisation was seen to block forever under rocprofv3)
    ex = None
    engine = local_csr = None
    in_csr = None
    import gc

    gc.collect()
    torch.cuda.synchronize()
    sys.stdout.flush()


if __name__ == "__main__":
    main()
