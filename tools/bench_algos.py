#!/usr/bin/env python3
"""WCC / SSSP / triangle count at BASELINE.json's configs (RMAT scale-22 WCC, scale-24 weighted SSSP and
triangle count): device time, parity against the oracle AT FULL SIZE, SURVEY §8(d)'s byte model -> roofline
fraction, and the oracle's time on this box's host cores as cpu_baseline.  One JSON object on stdout.

    python tools/bench_algos.py                    everything (the oracle runs take a few minutes of CPU)
    python tools/bench_algos.py --profile 1        one device call per algorithm, no oracle (for rocprofv3 passes)

Byte models (u32 ids, f32 weights / distances):
    WCC   4(n+1) + 4(m_out + m_in) + 8n                 both CSRs once + parent init / write-back
    SSSP  12 B x relaxed edges + 4 B x reached nodes    relaxed edges = out-edges of reached nodes, each counted once
    TC    4 B x wedges + 4 B x CSR entries + 8n   wedges = sum over DAG entries (u, v), v < u, of the rank of v in L(u);
          SURVEY 8(d)'s merge model (rank + |L(v)| per entry) is reported beside it
`roofline.frac` uses the wall time of the whole call (allocation, scheduling and the result download
included) — the kernel-only sums are in profiles/r02_algos_kernel_stats.txt.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--wcc-scale", type=int, default=22)
    ap.add_argument("--prapi-scale", type=int, default=0, help="scale of the page_rank() drop-in call timing (0: --wcc-scale)")
    ap.add_argument("--sssp-scale", type=int, default=24)
    ap.add_argument("--tc-scale", type=int, default=24)
    ap.add_argument("--oracle", type=int, default=1, help="1: the sequential checkers + the threaded timed legs (minutes of CPU); "
                    "2: only the threaded timed legs of WCC / SSSP (~2 s; their output is the parity bit), none for triangle "
                    "count (24 s; its equality with the oracle is asserted by tests/test_gpu_fullsize.py); 0: none")
    ap.add_argument("--tc-oracle", type=int, default=0, help="with --oracle 2: run orc_triangle_count as well (24 s on 16 cores at "
                    "scale 24), so that the count is checked in THIS process and the CPU leg is timed beside it")
    ap.add_argument("--profile", type=int, default=0)
    ap.add_argument("--skip", default="")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args(argv)
    if args.profile:
        args.oracle, args.reps = 0, 1
        args.skip += ",prapi"
    return args


def main():
    print(json.dumps(measure(parse())), flush=True)


def measure(args):
    """the legs above on this process's GPU; returns the record (bench.py calls this with --oracle 2 for its `extra` object)"""
    import numpy as np
    import torch

    from graph_amd import synth
    from graph_amd import prelude as P

    O = None
    cores = 0
    if args.oracle:
        from oracle import oracle as O  # checker + timed CPU baseline (oracle/graph_oracle.c header)

        cores = O.effective_cores()
    out = {"tool": "bench_algos", "device": torch.cuda.get_device_properties(0).name,
           "cpu_build": O.timed_build_flags() if O is not None else None}

    segments = []  # --profile: the label of every timed call, in launch order (tools/algos_profile.py cuts the trace there)

    def timed(fn, reps=None, label=None):
        """the reference app's protocol (crates/app/src/app.rs:124-153): the MEAN of the timed runs (warm-up runs are the
        caller's: the first / plan-building calls are timed on their own); the best run is kept beside it (timed.best)"""
        total, best, res = 0.0, None, None
        reps = reps or args.reps
        for _ in range(reps):
            res = None  # the previous result goes back to the (pinned) host allocator's cache before the next call
            if args.profile:
                torch.cuda._sleep(1000)  # marker dispatch (`spin_kernel`) in front of the call
                segments.append(label or f"call {len(segments)}")
            torch.cuda.synchronize()
            t = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            if args.profile:
                torch.cuda._sleep(1000)  # ... and behind it: what follows (the next graph's construction) is no part of the call
                segments.append("~")
            total += dt
            best = dt if best is None else min(best, dt)
        timed.best = best
        return total / reps, res

    # counter-measured HBM bytes per call (profiles/algos_traffic.json, tools/algos_traffic.py): quoted only when the record was
    # taken on the library this process loaded
    traffic_rec = {}
    try:
        import hashlib

        import graph_amd

        rec = json.load(open(os.path.join(ROOT, "profiles", "algos_traffic.json")))
        with open(graph_amd.LIB_PATH, "rb") as fh:
            if hashlib.sha256(fh.read()).hexdigest() == rec.get("library_sha256"):
                traffic_rec = rec
    except Exception:
        traffic_rec = {}

    def roofline(alg_bytes, seconds, key=None):
        ach = alg_bytes / seconds
        r = {"bound": "hbm", "algorithmic_bytes": int(alg_bytes), "achieved": round(ach / 1e9, 2), "peak": HBM_PEAK / 1e9,
             "unit": "GB/s", "frac": round(ach / HBM_PEAK, 5), "timed": "wall time of the whole API call", "traffic": None}
        t = traffic_rec.get(key) if key else None
        if t:
            r["traffic"] = t["hbm_bytes_per_call"]
            r["traffic_GBps"] = round(t["hbm_bytes_per_call"] / seconds / 1e9, 1)
            r["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summed over the call's kernels on this library "
                                   f"({traffic_rec['library_sha256'][:16]}), {traffic_rec.get('measured', '')}")
        else:
            r["traffic_source"] = "no counter record of this library (profiles/algos_traffic.json)"
        return r

    if "prapi" not in args.skip:
        # the drop-in call page_rank(&graph, config) with host result buffers: first call builds the
        # propagation-blocking plan (cached in the CSR handle), later calls reuse it
        sc = args.prapi_scale or args.wcc_scale
        n = 1 << sc
        src, dst = synth.rmat_edges(sc, 42)
        g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                               synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
        del src, dst
        cfg = P.PageRankConfig(20, 0.0, 0.85)
        t_first, _ = timed(lambda: P.page_rank(g, cfg), reps=1)
        t_next, res = timed(lambda: P.page_rank(g, cfg), reps=3)
        t_def, res_d = timed(lambda: P.page_rank(g, P.PageRankConfig()), reps=2)
        # (Auto = block-Gauss-Seidel sweeps on the propagation-blocking engine since round 6; the synchronous ones beside them)
        t_jac, res_j = timed(lambda: P.page_rank(g, cfg, P.PageRankMode.JacobiPB), reps=3)
        t_jdef, res_jd = timed(lambda: P.page_rank(g, P.PageRankConfig(), P.PageRankMode.JacobiPB), reps=2)
        out["page_rank_api"] = {"scale": sc, "first_call_ms": t_first * 1e3, "next_call_ms": t_next * 1e3,
                                "sweeps": res[1], "default_config_ms": t_def * 1e3, "default_config_sweeps": res_d[1],
                                "default_config_error": res_d[2], "synchronous_20_sweeps_ms": t_jac * 1e3,
                                "synchronous_default_config_ms": t_jdef * 1e3, "synchronous_default_config_sweeps": res_jd[1],
                                "synchronous_default_config_error": res_jd[2]}
        del g
        torch.cuda.empty_cache()

    if "wcc" not in args.skip:
        sc = args.wcc_scale
        n = 1 << sc
        src, dst = synth.rmat_edges(sc, 42)
        m = int(src.numel())
        g_out = synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted)
        g_in = synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted)
        g = P.DirectedCsrGraph(g_out, g_in, P.CsrLayout.Sorted)
        del src, dst
        t_wcc_first, _ = timed(lambda: P.wcc_afforest(g, P.WccConfig()).to_vec(), reps=1, label="wcc call 1 (allocates the parked buffers)")
        if not args.profile:  # (the reference's app: 5 warm-up runs, app.rs:66-72; here 1 + 2)
            timed(lambda: P.wcc_afforest(g, P.WccConfig()).to_vec(), reps=2, label="wcc warm-up")
        t_aff, comp = timed(lambda: P.wcc_afforest(g, P.WccConfig()).to_vec(), label="wcc steady")
        t_aff_best = timed.best
        rec = {"config": f"RMAT scale-{sc} DirectedCsrGraph<u32> wcc_afforest (labels = min id, what UndirectedCsrGraph "
                         f"wcc means, SURVEY a-5)", "nodes": n, "edges": m, "ms": t_aff * 1e3, "best_ms": t_aff_best * 1e3, "first_call_ms": t_wcc_first * 1e3,
               "edges_per_s": m / t_aff, "components": int(np.unique(comp).size)}
        rec["roofline"] = roofline(4 * (n + 1) + 4 * (2 * m) + 8 * n, t_aff, "wcc")
        if not args.profile:  # the same call with the ids LEFT ON THE DEVICE (components_out may be a device address)
            d_lab = torch.empty(n, dtype=torch.int32, device="cuda")
            t_dev, lab_dev = timed(lambda: P.wcc_afforest(g, P.WccConfig(), device_out=d_lab), label="wcc steady, device result")
            rec["ms_result_left_on_device"], rec["best_ms_result_left_on_device"] = t_dev * 1e3, timed.best * 1e3
            assert np.array_equal(lab_dev.cpu().numpy().view(np.uint32), comp)
        if not args.profile:
            t_base, comp_b = timed(lambda: P.wcc_baseline(g).to_vec())
            rec["baseline_ms"] = t_base * 1e3
            rec["afforest_eq_baseline"] = bool(np.array_equal(comp, comp_b))
        if O is not None:
            ooff, otgt, _ = g_out.host()
            ioff, itgt, _ = g_in.host()
            thr_out, cpu_s = O.wcc_afforest_timed(ooff, otgt, ioff, itgt, cores)  # the baseline: the reference's threading
            if args.oracle == 1:
                ref = O.wcc(ooff, otgt, ioff, itgt, O.AFFOREST, native=True)  # the checker: sequential
                rec["parity"] = {"bit_exact_vs_oracle": bool(np.array_equal(ref, comp)), "oracle": "orc_wcc AFFOREST (wcc.rs:158-301)",
                                 "threaded_baseline_eq_oracle": bool(np.array_equal(thr_out, ref))}
            else:
                rec["parity"] = {"bit_exact_vs_oracle": bool(np.array_equal(thr_out, comp)),
                                 "oracle": "orc_wcc_afforest_timed (the threaded restatement of wcc.rs:186-301; ids are the component minima "
                                           "under every schedule)"}
            rec["cpu_baseline"] = {"value": m / cpu_s, "unit": "edges/s", "seconds": cpu_s, "cores": cores, "kind": "port",
                                   "sample": "one full run of orc_wcc_afforest_timed on the same graph: 16384-node chunks from an "
                                             "atomic cursor, CAS union, parallel compress (wcc.rs:186-301, afforest.rs:22-53)"}
        out["wcc"] = rec
        del g, g_out, g_in
        torch.cuda.empty_cache()

    if "sssp" not in args.skip:
        sc = args.sssp_scale
        n = 1 << sc
        src, dst = synth.rmat_edges(sc, 42)
        m = int(src.numel())
        w = synth.rmat_weights(m, 44)
        g_out = synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted, w)
        del src, dst, w
        g = P.DirectedCsrGraph(g_out, g_out, P.CsrLayout.Sorted)
        deg = g_out.degrees()
        start = int(np.flatnonzero(deg > 0)[0])
        t_first, _ = timed(lambda: P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1)), reps=1, label="sssp call 1 (CSR lists)")  # allocates the scratch
        # the second call on a handle orders the lists by weight and transposes them (kept in the handle), then runs on them
        t_plan, _ = timed(lambda: P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1)), reps=1, label="sssp call 2 (builds the ordered lists)")
        t_s, dist = timed(lambda: P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1)), label="sssp steady (call 3+)")
        t_s_best = timed.best
        # the same call with the result LEFT ON THE DEVICE (gm_sssp_delta_stepping takes a device address as well): what the call
        # costs a caller that goes on working on the GPU — no n * 4 bytes over PCIe
        t_dev, dist_dev, t_dev_best = None, None, None
        if not args.profile:
            d_out = torch.empty(n, dtype=torch.float32, device="cuda")
            t_dev, dist_dev = timed(lambda: P.delta_stepping(g, P.DeltaSteppingConfig(start, 0.1), device_out=d_out), label="sssp steady, device result")
            t_dev_best = timed.best
            assert np.array_equal(dist_dev.cpu().numpy().view(np.uint32), dist.view(np.uint32))
        reached = dist < np.float32(3.0e38)
        relaxed = int(deg[reached].astype(np.int64).sum())
        rec = {"config": f"RMAT scale-{sc}, f32 weights uniform (0,1] seed 44, delta 0.1, start node {start}", "nodes": n,
               "edges": m, "ms": t_s * 1e3, "best_ms": t_s_best * 1e3, "ms_result_left_on_device": t_dev * 1e3 if t_dev else None,
               "best_ms_result_left_on_device": t_dev_best * 1e3 if t_dev_best else None, "first_call_ms": t_first * 1e3, "second_call_ms_builds_the_ordered_lists": t_plan * 1e3, "reached": int(reached.sum()), "relaxed_edges": relaxed,
               "relaxed_edges_per_s": relaxed / t_s}
        rec["roofline"] = roofline(12 * relaxed + 4 * int(reached.sum()), t_s, "sssp")
        if O is not None:
            off, tgt, wv = g_out.host()
            thr_out, cpu_s = O.delta_stepping_timed(off, tgt, wv, start, 0.1, cores)  # the baseline: thread-local bins
            if args.oracle == 1:
                ref = O.delta_stepping(off, tgt, wv, start, 0.1, native=True)  # the checker: sequential
                mis = O.stale_check_misfires(ref, 0.1)
                neq = int((ref.view(np.uint32) != dist.view(np.uint32)).sum())
                rec["parity"] = {"bit_exact_vs_oracle": neq == 0, "nodes_differing": neq,
                                 "stale_check_misfire_candidates": int(mis.sum()),
                                 "oracle": "orc_delta_stepping (sssp.rs:38-204)"}
                rec["parity"]["threaded_baseline_differs_on"] = int((thr_out.view(np.uint32) != ref.view(np.uint32)).sum())
            else:
                neq = int((thr_out.view(np.uint32) != dist.view(np.uint32)).sum())
                rec["parity"] = {"bit_exact_vs_oracle": neq == 0, "nodes_differing": neq,
                                 "oracle": "orc_delta_stepping_timed (the threaded restatement of sssp.rs:64-204)"}
            rec["cpu_baseline"] = {"value": relaxed / cpu_s, "unit": "relaxed edges/s", "seconds": cpu_s, "cores": cores,
                                   "kind": "port", "sample": "one full run of orc_delta_stepping_timed on the same graph: one set of "
                                                             "bins per thread, 64-node batches of the shared frontier, CAS on the "
                                                             "distances (sssp.rs:64-204)"}
        out["sssp"] = rec
        del g, g_out
        torch.cuda.empty_cache()

    if "tc" not in args.skip:
        sc = args.tc_scale
        n = 1 << sc
        src, dst = synth.rmat_edges(sc, 42)
        t0 = time.perf_counter()
        ug = P.UndirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Undirected, P.CsrLayout.Deduplicated),
                                  P.CsrLayout.Deduplicated)
        t_build = time.perf_counter() - t0
        del src, dst
        t0 = time.perf_counter()
        P.relabel_graph(ug)
        torch.cuda.synchronize()
        t_relabel = time.perf_counter() - t0
        t_tc_first, tri_first = timed(lambda: P.global_triangle_count(ug), reps=1, label="tc call 1 (builds the DAG and the list records)")  # builds the DAG + list records, kept in the handle
        t_tc, tri = timed(lambda: P.global_triangle_count(ug), reps=max(args.reps, 1), label="tc steady")
        t_tc_best = timed.best
        assert tri == tri_first
        rec = {"config": f"RMAT scale-{sc} to_undirected(Deduplicated) + make_degree_ordered (the --relabel path)",
               "nodes": n, "undirected_entries": ug.csr.m, "build_s": t_build, "relabel_s": t_relabel, "ms": t_tc * 1e3, "best_ms": t_tc_best * 1e3,
               "first_call_ms": t_tc_first * 1e3,
               "triangles": tri, "edges_per_s": ug.csr.m / 2 / t_tc, "triangles_per_s": tri / t_tc}
        off, tgt, _ = ug.csr.host()
        # SURVEY 8(d)'s merge-stream model: for every entry v < u of N(u): (rank of v in L(u)) + |L(v)| elements.
        # The device algorithm (bit row of L(v) in LDS, fronts of L(u) streamed) only needs the first term — the
        # wedges — plus one pass over the lists that name the pairs.
        d_off = torch.from_numpy(off.astype(np.int64)).cuda()
        d_tgt = torch.from_numpy(tgt.astype(np.int64)).cuda()
        rows = torch.repeat_interleave(torch.arange(n, device="cuda"), d_off[1:] - d_off[:-1])
        lower = d_tgt < rows
        low_len = torch.zeros(n, dtype=torch.int64, device="cuda").index_add_(0, rows[lower], torch.ones_like(rows[lower]))
        pos = torch.arange(d_tgt.numel(), device="cuda") - d_off[rows]
        wedges = int(pos[lower].sum().item())
        other = int(low_len[d_tgt[lower]].sum().item())
        dag_entries = int(lower.sum().item())
        del d_off, d_tgt, rows, lower, low_len, pos
        torch.cuda.empty_cache()
        rec["wedges"] = wedges
        rec["roofline"] = roofline(4 * wedges + 4 * ug.csr.m + 8 * n, t_tc, "tc")
        # No byte model is what this kernel reads (SURVEY 8(d)'s charges a second merge stream that is never read and comes out
        # above the peak; the wedge model below charges 4-byte ids where 94 % of the stream are 2-byte ids): `frac` is NOT
        # quoted for the triangle count.  What is measured is the counter traffic: traffic_GBps against the 8 TB/s peak.
        rec["roofline"]["frac_by_byte_model"] = rec["roofline"].pop("frac")
        rec["roofline"]["frac"] = (round(rec["roofline"]["traffic"] / t_tc / HBM_PEAK, 5) if rec["roofline"].get("traffic") else None)
        rec["roofline"]["frac_is"] = "counter-measured HBM bytes per call / wall time / 8 TB/s (null without a counter record of this library)"
        rec["roofline"]["bytes_model"] = ("4 B x wedges (sum over DAG entries (u, v) of the rank of v in L(u): the fronts of L(u) "
                                          "streamed against the bit row of L(v)) + 4 B x CSR entries + 8 B x nodes")
        rec["roofline"]["survey_merge_model"] = {"bytes": 4 * (wedges + other),
                                                 "achieved_GBps": round(4 * (wedges + other) / t_tc / 1e9, 1),
                                                 "note": "SURVEY 8(d): 4 B x sum (rank of v in L(u) + |L(v)|), the two streams of "
                                                         "the reference's sorted merge; the second term is not read here"}
        rec["dag_entries"] = dag_entries
        if O is not None and args.oracle != 1 and not args.tc_oracle:
            rec["parity"] = {"bit_exact_vs_oracle": None, "note": "not re-run here (24 s of host time): equality with orc_triangle_count "
                                                                  "at this size is asserted by tests/test_gpu_fullsize.py::"
                                                                  "test_scale24_triangle_count_equals_oracle; expected 10279340878 at scale 24",
                             "equals_the_count_pinned_by_that_test": bool(tri == 10279340878) if sc == 24 else None}
        if O is not None and (args.oracle == 1 or args.tc_oracle):
            t = time.perf_counter()
            ref = O.triangle_count(off, tgt, cores, native=True)
            cpu_s = time.perf_counter() - t
            rec["parity"] = {"bit_exact_vs_oracle": bool(ref == tri), "oracle_triangles": int(ref),
                             "oracle": "orc_triangle_count (triangle_count.rs:47-70)"}
            rec["cpu_baseline"] = {"value": ug.csr.m / 2 / cpu_s, "unit": "edges/s", "seconds": cpu_s, "cores": cores,
                                   "kind": "port", "sample": "one full run of orc_triangle_count (64-node dynamic chunks) "
                                                             "on the same relabelled graph"}
        out["tc"] = rec
        del ug
        torch.cuda.empty_cache()
    if args.profile:
        out["profile_segments"] = segments
    out["protocol"] = (f"ms = MEAN of {args.reps} timed API calls after the warm-up calls named beside it (crates/app/src/app.rs:124-153), "
                       "best_ms = the fastest of them")
    return out


if __name__ == "__main__":
    main()
