#!/usr/bin/env python3
"""Front (b) — one process per rank, bench.py's construction and PiecewiseExchange — with all ranks on ONE GPU over gloo,
against the single engine on the whole graph: the summed sweep error after EVERY sweep and the final scores, row by row.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29533 \
        tools/debug_multi_gloo.py --scale 20 --sweeps 25"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--sweeps", type=int, default=25)
ap.add_argument("--streams", type=int, default=0, help="(round 6: the stream-per-part schedule is gone; only 0 is accepted)")
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--sync", type=int, default=0, help="1: torch.cuda.synchronize() + barrier after every sweep")
ap.add_argument("--snap", type=int, default=0, help="1: keep the scores of every sweep (a copy per sweep on the caller's stream: changes the timing)")
ap.add_argument("--gather", choices=["async", "blocking", "main"], default="async",
                help="blocking: wait for a region's all-gather as soon as it is issued; main: issue every all-gather from the caller's stream")
args = ap.parse_args()

from graph_amd import synth
from graph_amd._lib import check, lib, vp
from graph_amd.distributed import PiecewiseExchange, rank_local_rows, split_exchange_layout
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, DeviceCsr, Direction

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
scale, n = args.scale, 1 << args.scale

in_csr, bounds, out_deg, _ = rank_local_rows(scale, 42, rank, world, 0, 16, collective=True)
row_lo, row_hi = int(bounds[rank]), int(bounds[rank + 1])
n_local = row_hi - row_lo
layout = split_exchange_layout(out_deg, bounds, parts=args.parts)
h = vp()
check(lib().gm_csr_slice_rows_map(in_csr.handle, row_lo, row_hi, layout["node_map"].data_ptr(), C.byref(h)))
local_csr = DeviceCsr(h)
from graph_amd.distributed import source_flags
local_csr.set_source_flags(source_flags(layout["node_map"], in_csr.no_in_edges, layout["x_len"]))
out_deg_local = out_deg[row_lo:row_hi].contiguous() if n_local else torch.zeros(1, dtype=torch.int32, device=dev)
engine = PageRankEngine(local_csr.handle, n, row_lo, out_deg_local, 0.85, x_len=layout["x_len"], engine=2)
scores = torch.zeros(max(n_local, 1), dtype=torch.float32, device=dev)
err = torch.zeros(1, dtype=torch.float64, device=dev)
assert args.streams == 0, "PiecewiseExchange(streams=True) was removed in round 6 (profiles/r06_streams_repro.txt)"
ex = PiecewiseExchange(engine, layout, rank, n_local, dev)
if args.gather != "async":
    orig, main_stream = ex._start_gather, torch.cuda.current_stream()

    def patched(buf, k):
        if args.gather == "blocking":
            orig(buf, k)
            if ex.works[k] is not None:
                ex.works[k].wait()
            return
        here = torch.cuda.current_stream()
        if here == main_stream:
            return orig(buf, k)
        ev = torch.cuda.Event(); ev.record(here); main_stream.wait_event(ev)
        with torch.cuda.stream(main_stream):
            orig(buf, k)
            ev2 = torch.cuda.Event(); ev2.record(main_stream)
        here.wait_event(ev2)

    ex._start_gather = patched
ex.start(scores)
errs, snaps = [], []
for t in range(args.sweeps):
    ex.sweep(scores, err)
    if args.snap or t == args.sweeps - 1:
        snaps.append(scores[:n_local].clone())  # (on the caller's stream, behind the sweep: main waits for every part)
    if args.sync:
        torch.cuda.synchronize()
        dist.barrier()
    e = err.detach().cpu().clone()
    dist.all_reduce(e, op=dist.ReduceOp.SUM)
    errs.append(float(e.item()))
ex.finish()
torch.cuda.synchronize()
mine = torch.stack(snaps).cpu() if n_local else torch.empty(len(snaps), 0)
sizes = [int(bounds[p + 1]) - int(bounds[p]) for p in range(world)]
parts = [torch.empty(len(snaps), s, dtype=torch.float32) for s in sizes] if rank == 0 else None
if rank == 0:
    parts[0].copy_(mine)
    for p in range(1, world):
        dist.recv(parts[p], src=p)
else:
    dist.send(mine, dst=0)
dist.barrier()
if rank == 0:
    multi_all = torch.cat(parts, dim=1).numpy()
    multi = multi_all[-1]
    src, dst = synth.rmat_edges(scale, 42, 16, 0)
    od = torch.bincount(src, minlength=n).to(torch.int32)
    whole = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted, None, 0)
    del src, dst
    one = PageRankEngine(whole.handle, n, 0, od, 0.85, x_len=n, engine=2)
    s1 = torch.zeros(n, dtype=torch.float32, device=dev)
    e1 = torch.zeros(1, dtype=torch.float64, device=dev)
    x = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(2)]
    one.init(s1, x[0])
    errs1, cur, first_bad = [], 0, None
    off = np.empty(n + 1, np.uint32)
    check(lib().gm_csr_download(whole.handle, off.ctypes.data_as(vp), None, None))
    deg = np.diff(off.astype(np.int64))
    bnd = np.asarray(bounds.cpu() if hasattr(bounds, "cpu") else bounds).astype(np.int64)
    for t in range(args.sweeps):
        one.sweep_tiles(x[cur], x[1 - cur], s1)
        one.sweep_fixup(x[1 - cur], s1, e1)
        cur = 1 - cur
        errs1.append(float(e1.item()))
        if first_bad is None and args.snap:
            ref = s1.cpu().numpy()
            d = np.nonzero(multi_all[t] != ref)[0]
            if d.size:
                own = np.searchsorted(bnd[1:], d, side="right")
                splits = [layout["row_splits"][p] for p in range(world)]
                grp = [int(np.searchsorted(np.asarray(splits[o][1:-1], dtype=np.int64), r - bnd[o], side="right")) for o, r in zip(own[:200000], d[:200000])]
                first_bad = {"sweep": t, "rows": int(d.size), "hub_rows": int((deg[d] >= 4096).sum()), "hub_rows_in_graph": int((deg >= 4096).sum()),
                             "by_owner_rank": np.bincount(own, minlength=world).tolist(), "by_row_group_of_owner": np.bincount(grp, minlength=args.parts).tolist(),
                             "max_rel": float(np.max(np.abs(multi_all[t][d].astype(np.float64) - ref[d]) / ref[d])),
                             "in_degree_min_med_max": [int(deg[d].min()), int(np.median(deg[d])), int(deg[d].max())],
                             "sample": [(int(r), int(deg[r]), float(multi_all[t][r]), float(ref[r])) for r in d[:6]]}
    single = s1.cpu().numpy()
    diff = np.nonzero(multi != single)[0]
    first_err = next((t for t in range(args.sweeps) if errs[t] != errs1[t]), None)
    rel = [abs(a - b) / b for a, b in zip(errs, errs1)]
    owner = np.searchsorted(np.asarray(bounds.cpu() if hasattr(bounds, "cpu") else bounds)[1:], diff, side="right") if diff.size else []
    print(json.dumps({
        "scale": scale, "world": world, "streams": args.streams, "sync": args.sync, "snap": args.snap, "gather": args.gather, "sweeps": args.sweeps,
        "rows_that_differ": int(diff.size), "of": n,
        "hub_rows_among_them": int((deg[diff] >= 4096).sum()) if diff.size else 0,
        "max_rel_score_diff": float(np.max(np.abs(multi[diff].astype(np.float64) - single[diff]) / single[diff])) if diff.size else 0.0,
        "by_owner_rank": np.bincount(owner, minlength=world).tolist() if diff.size else [],
        "first_sweep_whose_error_differs": first_err, "first_sweep_whose_scores_differ": first_bad, "env": {k: v for k, v in os.environ.items() if k.startswith("GM_")}, "max_rel_error_diff": max(rel),
        "errors_last": [errs[-1], errs1[-1]], "errors_first3": [errs[:3], errs1[:3]],
    }), flush=True)
dist.destroy_process_group()
