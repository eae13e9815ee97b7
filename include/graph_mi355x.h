/*
 * graph_mi355x.h — C ABI of the MI355X-native hot path behind neo4j-labs/graph's `graph::prelude`.
 *
 * The reference (Rust) has no FFI for its algorithms: they are generic free functions
 * re-exported from crates/algos/src/prelude.rs:1-7.  This header is the boundary a Rust
 * `extern "C"` shim (bindings/rust/) binds so that
 *   page_rank / wcc_afforest / delta_stepping / global_triangle_count
 * run on a CSR that was uploaded to HBM once.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns GM_OK (0) or a negative gm_status; nothing aborts across the ABI.
 *     gm_last_error() gives a thread-local message for the last failure on this thread.
 *   - node ids on the device are u32 (the reference's NI = u32; usize graphs are narrowed at
 *     upload with a range check -> GM_ERR_RANGE).
 *   - host buffers are owned by the caller; the graph a gm_csr holds is immutable after creation and
 *     the handle may be used from several host threads at once.
 *   - WHAT A HANDLE RETAINS.  Besides the graph, a handle keeps what its algorithms derived from it
 *     or last worked in, so that repeated calls (the reference's app times each algorithm in a loop)
 *     do not pay for it again: PageRank's propagation-blocking plan (~4.3 B/edge) and the stream,
 *     vectors and engine of the last gm_page_rank call (~3.4 B/edge + 12 B/node: ~9 GB at RMAT
 *     scale 26); the working set of the last gm_sssp_delta_stepping (~9 B/node; from the second call on a graph of
 *     2^20 edges or more also its lists ordered by weight and transposed, 16 B/edge + 4 B/node) and gm_wcc_* call
 *     (~8 B/node); gm_triangle_count's DAG of lower prefixes and list records (~6 B/entry +
 *     128 B/node: 4.7 GB at scale 24); the partition of the last gm_page_rank_multi call (slices,
 *     engines, exchange buffers on every device it named).  A concurrent second call of one algorithm allocates its own
 *     working set.  gm_csr_trim() releases all of it (the next call rebuilds what it needs);
 *     gm_csr_free() releases everything.  Large buffers come from a per-device arena of 64 MiB
 *     physical pieces that the library keeps for reuse (up to GM_ARENA_KEEP_GIB, default 32);
 *     gm_trim() returns the unused ones to the driver.
 *   - "device pointer" arguments are plain addresses in the HBM of the handle's device
 *     (e.g. hipMalloc or torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*.
 */
#ifndef GRAPH_MI355X_H
#define GRAPH_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_ABI_VERSION 1

typedef enum gm_status {
    GM_OK = 0,
    GM_ERR_INVALID = -1,     /* bad argument (null pointer, n == 0 where the reference panics, ...) */
    GM_ERR_RANGE = -2,       /* id / count does not fit the device id type, start_node >= n */
    GM_ERR_HIP = -3,         /* a HIP runtime call failed (message has the hipError string) */
    GM_ERR_NOMEM = -4,
    GM_ERR_UNSUPPORTED = -5  /* e.g. unsorted lists handed to triangle count */
} gm_status;

/* crates/builder/src/graph/csr.rs:34-45 */
typedef enum gm_layout { GM_LAYOUT_UNSORTED = 0, GM_LAYOUT_SORTED = 1, GM_LAYOUT_DEDUPLICATED = 2 } gm_layout;
/* crates/builder/src/lib.rs Direction (Outgoing / Incoming / Undirected) */
typedef enum gm_direction { GM_DIR_OUTGOING = 0, GM_DIR_INCOMING = 1, GM_DIR_UNDIRECTED = 2 } gm_direction;

const char *gm_last_error(void);
int gm_abi_version(void);
int gm_device_count(int *count_out);

/* ---------------------------------------------------------------------------------------------
 * Device-resident CSR.  Replaces the host `Csr<Index, NI, EV>{offsets, targets}` of
 * crates/builder/src/graph/csr.rs:58-118 as the thing the algorithms read; one gm_csr per
 * direction (DirectedCsrGraph = csr_out + csr_inc, csr.rs:364-368; UndirectedCsrGraph = csr,
 * csr.rs:658-661).  `weights` (optional) are the f32 edge values of Target<NI, f32>
 * (crates/builder/src/graph/mod.rs:5-10), stored SoA on the device.
 * ------------------------------------------------------------------------------------------- */
typedef struct gm_csr gm_csr;

int gm_csr_upload_u32(const uint32_t *offsets /* n+1 */, const uint32_t *targets /* m */,
                      const float *weights /* m or NULL */, uint64_t n, uint64_t m, int device,
                      gm_csr **out);
/* usize / u64 graphs: narrowed to u32, GM_ERR_RANGE when n or m >= 2^32 */
int gm_csr_upload_u64(const uint64_t *offsets, const uint64_t *targets, const float *weights, uint64_t n,
                      uint64_t m, int device, gm_csr **out);
/* Borrow arrays that already live in HBM (u32 offsets n+1, u32 targets m, f32 weights or 0).
 * The caller keeps them alive for the lifetime of the handle.  Like the uploads, the arrays are checked once
 * on the device (offsets ascending within [0, m] -> GM_ERR_INVALID, targets < n -> GM_ERR_RANGE). */
int gm_csr_wrap_device(uint64_t d_offsets, uint64_t d_targets, uint64_t d_weights, uint64_t n, uint64_t m,
                       int device, gm_csr **out);
void gm_csr_free(gm_csr *csr);
/* For a handle that holds a ROW SLICE of a partitioned graph whose lists index an exchange vector (gm_pr_create_with's x_len): one byte
 * per entry of that vector in device memory, non-zero = "the node in this slot has at most ONE in-edge".  A slice cannot see that in its own
 * offsets, and the propagation-blocking plan's rule for rows that sum many equal terms (GM_PB_HUB_LEAVES; the reference's
 * left-to-right sum of equal terms, page_rank.rs:143-146, drifts systematically: such rows are summed its way whatever their length)
 * needs it to flag the same rows as the single-GPU plan of the whole graph does.  The bytes are copied; plans built before the call are
 * dropped.  d_flags = 0 or len = 0 removes them.  Whole graphs need no flags: a source's in-degree is in the handle's own offsets. */
int gm_csr_set_source_flags(gm_csr *csr, uint64_t d_flags, uint64_t len);
/* Releases what the handle has parked for later calls (see "WHAT A HANDLE RETAINS" above): plans, DAGs,
 * working sets.  The graph itself stays; calls in flight keep what they are using. */
int gm_csr_trim(const gm_csr *csr);
/* Returns the arena's unused physical memory on `device` (-1: the current device) to the driver. */
int gm_trim(int device);
/* info_out[4]: bytes of device memory the arena holds, bytes of it not in use, 64 MiB pieces created so far,
 * pieces handed out so far. */
int gm_arena_info(int device, uint64_t *info_out);
/* info_out[3]: bytes of virtual address space the arena has reserved on `device` (never returned to the runtime: ROCm 7.0
 * corrupts later allocations after hipMemAddressFree of a range that carried mappings), bytes of it in released ranges
 * waiting for a request of their size class, bytes never handed out.  Buffer sizes come in classes (at most 12.5 % above
 * the request), so a process that builds graphs of many sizes reuses a bounded set of ranges. */
int gm_arena_va_info(int device, uint64_t *info_out);
uint64_t gm_csr_node_count(const gm_csr *csr);
uint64_t gm_csr_edge_count(const gm_csr *csr); /* number of target entries (csr.rs:76-78) */
int gm_csr_device(const gm_csr *csr);
uint64_t gm_csr_offsets_ptr(const gm_csr *csr); /* device addresses, for zero-copy consumers */
uint64_t gm_csr_targets_ptr(const gm_csr *csr);
uint64_t gm_csr_weights_ptr(const gm_csr *csr);
int gm_csr_download(const gm_csr *csr, uint32_t *offsets, uint32_t *targets, float *weights_or_null);
/* per-node degree = offsets[u+1]-offsets[u] (DirectedDegrees, crates/builder/src/lib.rs:330-340) */
int gm_csr_degrees(const gm_csr *csr, uint32_t *degrees_out /* n, host */);

/* Device-side CSR construction from an edge list in HBM — the sequential semantics of
 * `Csr::from((edges, node_count, direction, layout))`, crates/builder/src/graph/csr.rs:124-221
 * (+ sort_targets :886-895, sort_and_deduplicate_targets :897-948): Unsorted keeps edge-list
 * arrival order inside a list (out-direction entries first for Undirected), Sorted orders by
 * target (equal targets keep arrival order), Deduplicated also drops duplicates and self-loops.
 * d_src / d_dst: u32[m], d_weights: f32[m] or 0.  The inputs are not modified. */
int gm_csr_build_device(uint64_t n, uint64_t m, uint64_t d_src, uint64_t d_dst, uint64_t d_weights,
                        int direction, int layout, int device, gm_csr **out);
/* Same, from host edge arrays (uploads them, then builds on the device) — the device-resident
 * counterpart of GraphBuilder::new().csr_layout(l).edges(..).build(), crates/builder/src/builder.rs:123-540. */
int gm_csr_build_host(uint64_t n, uint64_t m, const uint32_t *src, const uint32_t *dst, const float *weights,
                      int direction, int layout, int device, gm_csr **out);
/* ToUndirectedOp::to_undirected(layout), crates/builder/src/graph_ops.rs:176-230 + csr.rs:391-464: an
 * Undirected build over the out-edges of a directed graph's out-CSR, on the device. */
int gm_csr_to_undirected(const gm_csr *out_csr, int layout, gm_csr **out);
/* RelabelByDegreeOp::make_degree_ordered, crates/builder/src/graph_ops.rs:511-638: returns a new
 * handle; new_id_out (host, n, optional) receives old id -> new id. */
int gm_csr_relabel_by_degree(const gm_csr *undirected, gm_csr **out, uint32_t *new_id_out);

/* Row slice for 1-D vertex-range partitioning: rows [row_lo, row_hi) of `full` with offsets
 * rebased to 0.  If `bounds` (host, parts+1 ascending node ids, bounds[0] = 0, bounds[parts] = n)
 * is given, every target id v in part p is rewritten to p*stride + (v - bounds[p]) — the index
 * of v in a rank-major all-gather buffer whose per-rank slot is `stride` floats — so the gathered
 * vector can be consumed without unpacking.  bounds == NULL keeps target ids. */
int gm_csr_slice_rows(const gm_csr *full, uint64_t row_lo, uint64_t row_hi, const uint32_t *bounds,
                      uint32_t parts, uint32_t stride, gm_csr **out);

/* Same, with an arbitrary rewrite of the target ids: d_map is a device array u32[n], every target v of
 * the slice becomes d_map[v].  Used to exchange only the out_scores of nodes that HAVE out-edges
 * (the others are never gathered): the map sends node v to its slot in the compacted rank-major
 * all-gather buffer (graph_amd/distributed.py:compact_exchange_layout). */
int gm_csr_slice_rows_map(const gm_csr *full, uint64_t row_lo, uint64_t row_hi, uint64_t d_map, gm_csr **out);

/* ---------------------------------------------------------------------------------------------
 * PageRank — replaces `page_rank(&G, PageRankConfig) -> (Vec<f32>, usize, f64)`,
 * crates/algos/src/page_rank.rs:58-111 (config :14-56: max_iterations 20, tolerance 1e-4,
 * damping_factor 0.85).  `in_csr` is the Incoming CSR (in_neighbors), `out_degree` the
 * per-node out-degrees (host, n; NULL = derive them on the device by counting each id's
 * occurrences in the in-lists, which equals out_degree for a DirectedCsrGraph).
 *
 * mode GM_PR_AUTO   : n <= 16384 -> GM_PR_SEQUENTIAL, else GM_PR_JACOBI (as GM_PR_BLOCK_GS where the engine is PB)
 *      GM_PR_JACOBI : synchronous sweeps (double-buffered out_scores), deterministic
 *      GM_PR_SEQUENTIAL : the reference's exact in-place ascending-u order on one wavefront;
 *                     bit-exact with the reference wherever the reference itself is
 *                     deterministic (n <= CHUNK_SIZE = 16384, page_rank.rs:12,135)
 * Stop rule as page_rank.rs:105-109: iteration += 1; stop if error < tolerance ||
 * iteration == max_iterations (so at least one sweep always runs).
 * The in-CSR's handle keeps what a call builds and allocates for the next call on the same graph (the
 * propagation-blocking plan, the call's stream, vectors and engine: ~9 GB beside a 4.3 GB CSR at scale 26);
 * gm_csr_free releases it.
 * ------------------------------------------------------------------------------------------- */
typedef enum gm_pr_mode {
    GM_PR_AUTO = 0,
    GM_PR_JACOBI = 1,      /* synchronous sweeps, engine chosen by size (gm_pr_engine_kind below) */
    GM_PR_SEQUENTIAL = 2,
    GM_PR_JACOBI_PULL = 3, /* synchronous sweeps, force the pull-tile engine */
    GM_PR_JACOBI_PB = 4,   /* synchronous sweeps, force the propagation-blocking engine */
    GM_PR_JACOBI_REFORDER = 5, /* synchronous sweeps whose row sums are added left to right in f32 in CSR
                                 order, exactly like page_rank.rs:143-146 (one lane per row: a parity
                                 instrument, not a fast path) */
    GM_PR_BLOCK_GS = 6 /* block-Gauss-Seidel sweeps on the propagation-blocking engine: K row blocks (GM_PR_BLOCK_GS=K,
                          default 8) in ascending order, block j sees this sweep's out_scores of the blocks before it —
                          the reference's in-place update (page_rank.rs:142-160) at block granularity, deterministic.
                          About half the iterations of the synchronous sweeps for the same error; same fixed point.
                          GM_PR_AUTO runs these whenever it picks the propagation-blocking engine (GM_PR_BLOCK_GS=0:
                          synchronous sweeps instead). */
} gm_pr_mode;

int gm_page_rank(const gm_csr *in_csr, const uint32_t *out_degree, uint64_t max_iterations, double tolerance,
                 float damping_factor, int mode, float *scores_out /* n, host */, uint64_t *iterations_out,
                 double *error_out);
/* The same for a caller that holds both CSRs of a DirectedCsrGraph on the device (the shape of the
 * reference's graph type, csr.rs:364-389): out-degrees come from out_csr's offsets on the device, so
 * nothing but the scores crosses PCIe (gm_page_rank uploads n * 4 bytes of out-degrees per call). */
int gm_page_rank_directed(const gm_csr *out_csr, const gm_csr *in_csr, uint64_t max_iterations, double tolerance,
                          float damping_factor, int mode, float *scores_out, uint64_t *iterations_out, double *error_out);

/* page_rank() on a graph split over the GPUs of one node (BASELINE north_star: 1-D vertex-range partition, RCCL
 * exchange of the rank vector over xGMI every iteration), driven by ONE host thread — the entry a Rust / C++ host
 * calls; nothing in it needs Python.  Rows are cut into n_devices contiguous ranges balanced by in-degree (the
 * reference's greedy partitioner, crates/builder/src/graph_ops.rs:431-439,479-509); every device gets its rows of
 * the in-CSR, a sweep engine and a replica of out_scores; per sweep: local sweep kernels -> ncclAllGather of the
 * out_scores of the nodes that have out-edges -> the f64 error partials summed in rank order.  Same stop rule and
 * results as gm_page_rank_directed — bit-identical: ordinary rows are exactly rounded sums, hub rows the reference's own
 * left-to-right sums (page_rank.rs:143-146); neither depends on the partition.
 * The exchange travels in K regions (GM_MULTI_PARTS, default 2) on streams of its own, under the work of the other
 * regions; the host synchronises only when the stop rule needs the error (tolerance > 0: every sweep; 0: once).  The
 * partition, slices, engines, streams and communicator of a call are parked in in_csr's handle: a second call with the
 * same device list builds nothing (gm_csr_trim releases them).
 * devices: n_devices device ordinals, or NULL for 0 .. n_devices-1.  A device named more than once gives
 * "virtual ranks" (the exchange then uses device-to-device copies instead of RCCL) — for exercising the
 * partitioned path on a single GPU.  librccl.so is loaded on first use; GM_ERR_UNSUPPORTED if it is missing. */
int gm_page_rank_multi(const gm_csr *out_csr, const gm_csr *in_csr, const int *devices, uint32_t n_devices,
                       uint64_t max_iterations, double tolerance, float damping_factor, float *scores_out /* n, host */,
                       uint64_t *iterations_out, double *error_out);
/* The same run from PIECES, for graphs that never sit whole on one device (north_star: "graphs larger than one GPU are 1-D
 * vertex-range partitioned"): in_slices[p] = the rows [bounds[p], bounds[p + 1]) of the in-CSR, resident on devices[p], with
 * n_local rows, offsets from 0 and targets as GLOBAL node ids (e.g. gm_csr_build_device over the edges whose destination lies
 * in the range, then gm_csr_slice_rows); d_out_degree_full[p] = device address, on devices[p], of u32[n]: the out-degree of
 * EVERY node (pieces built apart get it by summing their histograms: an all-reduce).  bounds: n_devices + 1 ascending row
 * bounds from 0 to n — the reference's in_degree_partition ranges (graph_ops.rs:431-439,479-509) reproduce gm_page_rank_multi
 * bit for bit.  Every rank derives the exchange layout on its own device; the inputs are copied, not consumed; nothing is
 * parked.  (graph_amd/distributed.py: partition_local_slices builds such pieces from the counter-based R-MAT generator.) */
int gm_page_rank_multi_slices(const gm_csr *const *in_slices, const uint64_t *bounds, const uint64_t *d_out_degree_full,
                              uint64_t n, const int *devices, uint32_t n_devices, uint64_t max_iterations, double tolerance,
                              float damping_factor, float *scores_out /* n, host */, uint64_t *iterations_out,
                              double *error_out);

/* Resident PageRank engine: the per-sweep hot loop (page_rank_iteration, page_rank.rs:113-168)
 * over rows [row_begin, row_begin + n_local) of a graph with n_global nodes.  One engine per
 * GPU; with n_local == n_global it is the single-GPU path.  All arrays are device pointers. */
typedef struct gm_pr gm_pr;
int gm_pr_create(const gm_csr *in_csr_rows /* n_local rows, targets are global ids */, uint64_t n_global,
                 uint64_t row_begin, uint64_t d_out_degree_local /* u32[n_local] */, float damping_factor,
                 gm_pr **out);
/* Same with explicit choices.  x_len: length of the x_in vector the sweeps will read (n_global, or
 * parts*stride when targets were rewritten by gm_csr_slice_rows).  engine:
 *   GM_PR_ENGINE_PULL  merge-tile pull sweep (gather out_scores through the caches)
 *   GM_PR_ENGINE_PB    propagation blocking: values binned by destination range, accumulated in LDS
 *                      as exact 64-bit fixed point — all HBM traffic sequential (large graphs)
 *   GM_PR_ENGINE_AUTO  PB from 2^24 edges up, else PULL */
typedef enum gm_pr_engine_kind {
    GM_PR_ENGINE_AUTO = 0,
    GM_PR_ENGINE_PULL = 1,
    GM_PR_ENGINE_PB = 2,
    GM_PR_ENGINE_REFORDER = 3 /* per-row left-to-right f32 sums (the reference's rounding); slow, for parity studies */
} gm_pr_engine_kind;
int gm_pr_create_with(const gm_csr *in_csr_rows, uint64_t n_global, uint64_t row_begin, uint64_t x_len,
                      uint64_t d_out_degree_local, float damping_factor, int engine, gm_pr **out);
int gm_pr_engine(const gm_pr *pr); /* the engine actually chosen */
void gm_pr_destroy(gm_pr *pr);
/* scores[i] = 1/n_global, x_local[i] = (1/n_global)/out_degree[i]   (page_rank.rs:70-81) */
int gm_pr_init(gm_pr *pr, uint64_t d_scores_local, uint64_t d_x_local, void *stream);
/* one synchronous sweep: reads x_in (f32[n_global], out_scores of the previous sweep), updates
 * scores_local in place, writes x_out_local (f32[n_local]) and this rank's share of the sweep's
 * L1 error to *d_error_out (f64, device).  Enqueues two kernels on `stream`; never synchronises.
 * Deterministic: no floating-point atomics anywhere. */
int gm_pr_sweep(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                uint64_t d_error_out, void *stream);
/* the two launches of gm_pr_sweep separately (bench.py brackets the dominant tile kernel with
 * its own HIP events): tiles = gather + per-row reduce + fused epilogue; fixup = rows that cross
 * a tile boundary + the deterministic error reduction. */
int gm_pr_sweep_tiles(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                      void *stream);
int gm_pr_sweep_fixup(gm_pr *pr, uint64_t d_x_out_local, uint64_t d_scores_local, uint64_t d_error_out,
                      void *stream);
/* A sweep in pieces, for partitioned runs that overlap the exchange of one part of out_scores with the
 * work on another (propagation-blocking engines only, GM_ERR_UNSUPPORTED otherwise; SURVEY §8e "must be
 * overlapped").  The caller lays the global vector out in regions that are multiples of the source tile
 * (gm_pr_part_geometry) and splits its rows at multiples of rows_per_bin; then per sweep:
 * gm_pr_sweep_bin for every region [x_lo, x_hi) of the vector as it arrives; once all of it is there the hot
 * sources are staged (gm_pr_sweep_hot, or stage_hot = 1 on the first accumulate) and gm_pr_sweep_accum runs
 * for every part, in order on one stream (what both fronts of this library do; parts on streams of their own, ordered by
 * events, were dropped in round 6: DESIGN.md section 6); gm_pr_sweep_fixup for the error.  Together they do exactly what gm_pr_sweep_tiles does: same kernels, same bits.
 * Hub rows (page_rank.rs:143-146 order) may lie in any part: their kernels are launched with part 0, beside its
 * accumulate kernel on the engine's own side streams, and gm_pr_sweep_accum of EVERY part makes its stream wait for
 * them behind its own accumulate kernel — what the caller enqueues next on that stream (the exchange of the part's
 * rows) sees the part's hub rows finished.  Enqueue part 0 first. */
int gm_pr_part_geometry(const gm_pr *pr, uint64_t *rows_per_bin_out, uint64_t *source_tile_out);
int gm_pr_set_parts(gm_pr *pr, const uint64_t *row_splits /* n_parts + 1 values: 0 .. n_local */, uint64_t n_parts);
int gm_pr_sweep_bin(gm_pr *pr, uint64_t d_x_in_global, uint64_t x_lo, uint64_t x_hi /* elements of x_in */, void *stream);
/* Regions given as lists of ranges: range i = [x_lo[i], x_hi[i]) (whole source tiles) belongs to region region[i];
 * gm_pr_sweep_bin_region propagates all ranges of one region in ONE launch.  A partitioned run keeps the exchanged
 * vector rank-major (ascending node ids — the order the hub rows' sums follow, page_rank.rs:143-146) and exchanges row
 * group k of every rank as region k: n_ranks ranges. */
int gm_pr_set_bin_regions(gm_pr *pr, const uint64_t *x_lo, const uint64_t *x_hi, const uint32_t *region, uint64_t count,
                          uint32_t n_regions);
int gm_pr_sweep_bin_region(gm_pr *pr, uint64_t d_x_in_global, uint32_t region, void *stream);
int gm_pr_sweep_hot(gm_pr *pr, uint64_t d_x_in_global, void *stream); /* stage the hot sources (whole vector needed) */
int gm_pr_sweep_accum(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                      uint64_t part, int stage_hot /* 1: gm_pr_sweep_hot first, on this stream */, void *stream);
/* algorithmic HBM bytes of one sweep, SURVEY §8(d): 8*m_local + 20*n_local + 4 */
uint64_t gm_pr_algorithmic_bytes(const gm_pr *pr);
uint64_t gm_pr_tile_count(const gm_pr *pr); /* workgroups per sweep (diagnostics) */
/* Diagnostics of the propagation-blocking plan behind an engine (GM_ERR_UNSUPPORTED for the other engines):
 * info[0] bytes of plan data in HBM, [1] time the plan took to build (microseconds; it is built once per graph
 * and cached in the gm_csr), [2] hub rows whose sums follow the reference's left-to-right f32 order
 * (page_rank.rs:143-146), [3] their in-edges, [4] the in-degree threshold for that (GM_PB_HUB_DEG, default
 * 4096, 0 = off), [5] hot sources, [6] entries of the value stream, [7] hot edges, [8] bytes of this engine's
 * scratch (value stream etc.), [9] bins, [10] source tiles, [11] (tile, bin) segments, [12] hub groups (the hub
 * rows are walked in groups of <= 64 rows, one workgroup each), [13] tiers of hot sources, [14] long rows (summed by
 * pb_hublong_kernel: a scan over parity pairs, a row of more than 8 passes over several workgroups), [15] their in-edges,
 * [16] 2040-term blocks of the other hub groups, [17] / [18] bin-kernel time (us) of the fastest / slowest timed
 * placement draw of the engine's value stream, [19] draws timed, [20] 64 MiB pieces the device arena was grown by for it,
 * [21] 1: the value stream is mapped from arena pieces (0: hipMalloc), [22] hub terms taken from hot sources (off the
 * value stream).  Further entries are 0.
 * A CSR whose hub rows' lists are NOT ascending (CsrLayout::Unsorted, csr.rs:34-45) is noticed when the plan is built:
 * its hub rows are summed in the order the lists lie in the CSR (the reference's, page_rank.rs:143-146) through a
 * per-term index (4 bytes per hub edge more plan data, an L2 gather per hub term and sweep). */
int gm_pr_plan_info(const gm_pr *pr, uint64_t *info, uint32_t count);

/* ---------------------------------------------------------------------------------------------
 * WCC — replaces wcc_afforest / wcc_afforest_dss / wcc_baseline(&G, WccConfig) -> impl
 * Components<NI>, crates/algos/src/wcc.rs:103-183 (config :43-79: chunk_size 16384 is a CPU
 * scheduling knob and has no device meaning, neighbor_rounds 2, sampling_size 1024).
 * components_out[u] = Components::component(u) = minimum node id of u's weakly connected
 * component — identical for all three reference entry points.
 * ------------------------------------------------------------------------------------------- */
int gm_wcc_afforest(const gm_csr *out_csr, const gm_csr *in_csr, uint64_t neighbor_rounds,
                    uint64_t sampling_size, uint32_t *components_out /* n, host */);
int gm_wcc_baseline(const gm_csr *out_csr, uint32_t *components_out);
/* Partitioned run (labels replicated on every GPU, u32[n_global] in HBM): gm_wcc_init_labels sets
 * label[i] = i; gm_wcc_link_rows links every edge of a rank's row slice(s) (row r of the slice is node
 * row_begin + r; targets are global ids; in_rows may be NULL) into the labels and compresses them.
 * Between rounds the ranks min-all-reduce the label vector (graph_amd/distributed.py:wcc_partitioned);
 * the fixed point is label[u] = minimum node id of u's component, as in the single-GPU path. */
int gm_wcc_init_labels(uint64_t n, uint64_t d_labels, int device, void *stream);
int gm_wcc_link_rows(const gm_csr *out_rows, const gm_csr *in_rows, uint64_t row_begin, uint64_t n_global,
                     uint64_t d_labels, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SSSP — replaces delta_stepping(&G, DeltaSteppingConfig{start_node, delta}) -> Vec<AtomicF32>,
 * crates/algos/src/sssp.rs:38-102.  `out_csr` must carry weights (>= 0).  Unreachable nodes get
 * f32::MAX (sssp.rs:12), not inf.  start_node >= n -> GM_ERR_RANGE (the reference panics, :52).
 * distances_out: n floats in host memory (page-locked memory takes the copy at link speed) or in device
 * memory.  The handle keeps the call's working buffers (~9 bytes per node + 0.3 bytes per edge) for the
 * next call and remembers that its weights passed the >= 0 / not-NaN check; the arrays of a wrapped CSR
 * must not change while the handle lives (the same rule as for PageRank's cached plan).
 * The SECOND call on a handle with >= 2^20 edges builds a plan (every list once more ordered by weight, and the lists
 * transposed: 16 bytes per edge + 4 per node, kept until gm_csr_trim; two radix sorts, ~36 bytes per edge while they
 * run) — only if the device has that much, and a quarter more, free; otherwise (and after a failed build, until
 * gm_csr_trim) calls keep running on the CSR's own lists.  One builder per handle; the distances do not depend on it.
 * ------------------------------------------------------------------------------------------- */
int gm_sssp_delta_stepping(const gm_csr *out_csr, uint64_t start_node, float delta, float *distances_out);
/* Partitioned run (distances replicated on every GPU as u32 bit patterns of non-negative f32, so an
 * integer min-all-reduce orders them): gm_sssp_init_distances fills f32::MAX and 0 at the start node;
 * gm_sssp_relax_rows relaxes every out-edge of a rank's weighted row slice whose source is reached and
 * ORs 1 into *d_changed (u32, device) when a distance improved.  graph_amd/distributed.py:
 * sssp_partitioned alternates local passes and min-all-reduces until nothing changes — the same least
 * fixed point as gm_sssp_delta_stepping. */
int gm_sssp_init_distances(uint64_t n, uint64_t start_node, uint64_t d_dist_bits, int device, void *stream);
int gm_sssp_relax_rows(const gm_csr *out_rows, uint64_t row_begin, uint64_t n_global, uint64_t d_dist_bits,
                       uint64_t d_changed, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Triangle count — replaces global_triangle_count(&G) -> u64, crates/algos/src/triangle_count.rs:22-86,
 * including its put-back-iterator semantics on lists with duplicates / self-loops
 * (crates/algos/src/utils.rs:8-101).  Lists must be sorted (layout Sorted or Deduplicated);
 * unsorted lists are rejected with GM_ERR_UNSUPPORTED (the reference returns garbage silently).
 * The handle keeps what the count derives from the graph alone (the lower-prefix DAG and its list records, ~9 bytes
 * per undirected entry + 128 bytes per node) for the next count on the same graph.
 * ------------------------------------------------------------------------------------------- */
int gm_triangle_count(const gm_csr *undirected_csr, uint64_t *triangles_out);

/* ---------------------------------------------------------------------------------------------
 * Synthetic inputs (no reference counterpart: the reference downloads Graph500 files,
 * crates/builder/benches/common/mod.rs:15-41).  R-MAT A=.57 B=.19 C=.19 D=.05, integer-only
 * arithmetic shared bit-for-bit with oracle/graph_oracle.c:orc_rmat_edges.
 * ------------------------------------------------------------------------------------------- */
int gm_rmat_edges_device(uint32_t scale, uint64_t seed, uint64_t first_edge, uint64_t count, uint64_t d_src,
                         uint64_t d_dst, int device, void *stream);
int gm_rmat_weights_device(uint64_t seed, uint64_t first_edge, uint64_t count, uint64_t d_weights, int device,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPH_MI355X_H */
