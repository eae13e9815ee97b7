// pagerank.hip — PageRank pull sweeps on a device-resident in-CSR (gfx950 / wave64).
//
// Replaces crates/algos/src/page_rank.rs:58-168 of the reference.  Per node u the arithmetic is
// the reference's, op for op in f32 with no contraction:
//     incoming = sum_{v in in_neighbors(u)} out_scores[v]        (page_rank.rs:143-146)
//     new      = base + damping * incoming                        (:149)
//     error   += |new - old| widened to f64                       (:152-153)
//     out_scores[u] = new / out_degree(u)                         (:155-159)
// What differs is scheduling.  The reference pulls 16384-node chunks with an atomic cursor and
// races on out_scores in place; here a sweep is synchronous (reads x_in, writes x_out) and its
// work unit is a *merge tile*: W consecutive items of the merged sequence
// "row marker r, then row r's in-edges" (position of row r's marker = off[r] + r).  Every
// tile therefore holds at most W edges AND at most W rows whatever the degree distribution
// (RMAT hubs with 10^5..10^6 in-edges, or millions of empty rows), so all workgroups stream
// the same number of bytes.
//
//   pr_tile_kernel   one 256-thread workgroup per tile: coalesced read of the tile's targets,
//                    up to 8 independent gathers of x_in per lane in flight, values parked in
//                    LDS, then one lane per row sums its LDS slice (rows longer than 32 edges
//                    are summed by a whole wavefront with a shuffle tree), fused epilogue
//                    (new score, |delta|, out_score).  A row that crosses a tile boundary leaves
//                    partial sums in tail[t] / head[t'].
//   pr_fixup_kernel  one lane per tile: finishes the (at most one) row that starts in the tile
//                    and crosses its end, sums the per-tile f64 errors in a fixed order (last
//                    block done reduces) -> deterministic results and error, no float atomics.
//
// Roofline: HBM.  Algorithmic bytes per sweep = 8m + 20n + 4: offsets 4(n+1),
// targets 4m, gathered out_scores 4m, old score 4n, new score 4n, out_score 4n, out-degree 4n.
//
// pr_seq_kernel is the reference's exact sequential order (one wavefront, out_scores in LDS):
// bit-exact with the reference wherever the reference is deterministic (n <= 16384).
#include "pagerank.hpp"

namespace {

using namespace gm;

constexpr int PR_BLOCK = 256;
constexpr int PR_WAVES = PR_BLOCK / kWave;
constexpr int PR_W = 2048;     // merged items per tile
constexpr int PR_EPT = PR_W / PR_BLOCK;
constexpr int PR_SHORT = 32;   // rows up to this many in-tile edges are summed by one lane
constexpr int PR_MAXLONG = PR_W / (PR_SHORT + 1) + 2;

// ---- setup: first row whose marker lies in tile t --------------------------------------------
// tile_row[t] = #rows r with off[r] + r < t*W  (t = 0..T-1), tile_row[T] = n.
__global__ void pr_tile_rows_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t T,
                                    uint32_t *__restrict__ tile_row)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > T)
        return;
    if (t == T) {
        tile_row[t] = n;
        return;
    }
    const uint64_t target = (uint64_t)t * PR_W;
    tile_row[t] = (uint32_t)lower_bound_fn(0, n, target, [&](uint64_t r) { return (uint64_t)off[r] + r; });
}

__global__ void pr_init_kernel(uint32_t n_local, float init, const uint32_t *__restrict__ outdeg,
                               float *__restrict__ scores, float *__restrict__ x)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_local; u += stride) {
        scores[u] = init;
        x[u] = __fdiv_rn(init, (float)outdeg[u]);
    }
}

// out_degree[v] = number of occurrences of v in the in-lists
__global__ void pr_degrees_from_offsets_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t *__restrict__ deg)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride)
        deg[u] = off[u + 1] - off[u];
}

// Is there a row that must be summed in the reference's order?  One of at least min_len entries — or (leaf_t != 0) a shorter one with at
// least leaf_t sources that have at most one in-edge themselves: many EQUAL terms, on which the reference's left-to-right sum drifts
// systematically (pagerank_pb.hip: pb_leafflag_kernel; whole graphs read a source's in-degree off the same offsets, a partition slice
// needs gm_csr_set_source_flags' bytes).  Rows between leaf_t and min_len entries are rare below 2^24 edges; a lane walks one.
__global__ void pr_long_row_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t n, uint32_t min_len,
                                   uint32_t leaf_t, const uint8_t *__restrict__ src_flags, uint64_t src_flags_len, int whole,
                                   uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    bool any = false;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n && !any; u += stride) {
        const uint32_t b = off[u], deg = off[u + 1] - b;
        any = deg >= min_len;
        if (!any && leaf_t && deg >= leaf_t && (whole || src_flags)) {
            uint32_t c = 0;
            for (uint32_t k = 0; k < deg; ++k) {
                const uint32_t s = src[b + k];
                c += src_flags ? ((s < src_flags_len && src_flags[s]) ? 1u : 0u) : ((s < n && off[s + 1] - off[s] <= 1u) ? 1u : 0u);
            }
            any = c >= leaf_t;
        }
    }
    if (any)
        *flag = 1u;
}

__global__ void pr_count_outdeg_kernel(const uint32_t *__restrict__ tgt, uint64_t m, uint32_t *__restrict__ outdeg)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        atomicAdd(&outdeg[tgt[i]], 1u);
}

// ---- the sweep ---------------------------------------------------------------------------------
struct Seg {
    uint32_t r;    // local row
    uint32_t s, e; // in-tile edge range, relative to the tile's first edge
    int kind;      // 0 = whole row (finalize), 1 = head partial, 2 = tail partial, -1 = nothing
};

__device__ __forceinline__ Seg pr_decode_seg(uint32_t j, uint32_t rs, uint32_t eb, uint32_t ee,
                                             const uint32_t *__restrict__ off)
{
    Seg g;
    if (j == 0) { // the row that started before this tile and still has edges here
        g.r = rs - 1;
        g.s = 0;
        const uint32_t end = off[rs]; // rs <= n_local: off has n_local + 1 entries
        g.e = (end < ee ? end : ee) - eb;
        g.kind = rs > 0 ? 1 : -1;
        if (rs == 0)
            g.e = 0;
        return g;
    }
    g.r = rs + j - 1;
    const uint32_t s = off[g.r], e = off[g.r + 1];
    g.s = s - eb; // s >= eb by construction of the tile
    if (e > ee) {
        g.e = ee - eb;
        g.kind = 2;
    } else {
        g.e = e - eb;
        g.kind = 0;
    }
    return g;
}

__global__ __launch_bounds__(PR_BLOCK) void pr_tile_kernel(
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const uint32_t *__restrict__ tile_row,
    const float *__restrict__ x_in, const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
    float *__restrict__ x_out, float *__restrict__ head, float *__restrict__ tail, double *__restrict__ tile_err,
    uint32_t m, float base, float damping)
{
    __shared__ float vals[PR_W];
    __shared__ uint32_t longseg[PR_MAXLONG];
    __shared__ uint32_t nlong;
    __shared__ float head_v, tail_v;
    __shared__ double red[PR_WAVES];

    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    const uint32_t rs = tile_row[t], re = tile_row[t + 1];
    const uint32_t eb = (uint32_t)((uint64_t)t * PR_W - rs);
    const uint64_t ee64 = (uint64_t)(t + 1) * PR_W - re;
    const uint32_t ee = ee64 < m ? (uint32_t)ee64 : m;
    const uint32_t ne = ee - eb;

    if (tid == 0) {
        nlong = 0;
        head_v = 0.0f;
        tail_v = 0.0f;
    }

    // phase 1: stream the tile's targets (coalesced), gather x_in, park in LDS
    {
        uint32_t v[PR_EPT];
#pragma unroll
        for (int k = 0; k < PR_EPT; ++k) {
            const uint32_t i = tid + k * PR_BLOCK;
            v[k] = i < ne ? tgt[eb + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < PR_EPT; ++k) {
            const uint32_t i = tid + k * PR_BLOCK;
            if (i < ne)
                vals[i] = x_in[v[k]];
        }
    }
    __syncthreads();

    // phase 2: one lane per row (segment 0 is the head partial)
    const uint32_t nseg = re - rs + 1;
    double err = 0.0;
    for (uint32_t j = tid; j < nseg; j += PR_BLOCK) {
        const Seg g = pr_decode_seg(j, rs, eb, ee, off);
        if (g.kind < 0)
            continue;
        const uint32_t len = g.e - g.s;
        if (len > PR_SHORT) {
            longseg[atomicAdd(&nlong, 1u)] = j;
            continue;
        }
        float sum = 0.0f;
        for (uint32_t i = g.s; i < g.e; ++i)
            sum += vals[i];
        if (g.kind == 0)
            err += pr_finalize(g.r, sum, base, damping, outdeg, scores, x_out);
        else if (g.kind == 1)
            head_v = sum;
        else
            tail_v = sum;
    }
    __syncthreads();

    // phase 3: long rows, one wavefront each
    const uint32_t nl = nlong;
    const uint32_t lane = tid & (kWave - 1), wave = tid >> 6;
    for (uint32_t k = wave; k < nl; k += PR_WAVES) {
        const Seg g = pr_decode_seg(longseg[k], rs, eb, ee, off);
        float sum = 0.0f;
        for (uint32_t i = g.s + lane; i < g.e; i += kWave)
            sum += vals[i];
        sum = wave_sum(sum);
        if (lane == 0) {
            if (g.kind == 0)
                err += pr_finalize(g.r, sum, base, damping, outdeg, scores, x_out);
            else if (g.kind == 1)
                head_v = sum;
            else
                tail_v = sum;
        }
    }
    const double total = block_sum<double, PR_WAVES>(err, red); // contains a __syncthreads()
    if (tid == 0) {
        tile_err[t] = total;
        head[t] = head_v;
        tail[t] = tail_v;
    }
}

// One lane per tile: finish the row that starts in tile t and ends in a later tile, then reduce
// the sweep's error deterministically (last block done sums the block partials in index order).
__global__ __launch_bounds__(PR_BLOCK) void pr_fixup_kernel(
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ tile_row, const uint32_t *__restrict__ outdeg,
    float *__restrict__ scores, float *__restrict__ x_out, const float *__restrict__ head,
    const float *__restrict__ tail, const double *__restrict__ tile_err, double *__restrict__ blk_err,
    uint32_t *__restrict__ ticket, double *__restrict__ err_out, uint32_t T, uint32_t m, float base, float damping)
{
    __shared__ double red[PR_WAVES];
    __shared__ bool is_last;
    const uint32_t tid = threadIdx.x;
    const uint32_t t = blockIdx.x * PR_BLOCK + tid;
    double err = 0.0;
    if (t < T) {
        err = tile_err[t];
        const uint32_t rs = tile_row[t], re = tile_row[t + 1];
        if (re > rs) {
            const uint32_t r = re - 1; // last row whose marker is in tile t
            const uint64_t ee64 = (uint64_t)(t + 1) * PR_W - re;
            const uint32_t ee = ee64 < m ? (uint32_t)ee64 : m;
            const uint32_t e = off[r + 1];
            if (e > ee) { // it crosses the tile end: its last edge sits at merged position e + r
                const uint32_t t1 = (uint32_t)(((uint64_t)e + r) / PR_W);
                float sum = tail[t];
                for (uint32_t k = t + 1; k <= t1; ++k)
                    sum += head[k];
                err += pr_finalize(r, sum, base, damping, outdeg, scores, x_out);
            }
        }
    }
    const double total = block_sum<double, PR_WAVES>(err, red);
    if (tid == 0) {
        st_agent(&blk_err[blockIdx.x], total);
        __threadfence();
        const uint32_t prev = atomicAdd(ticket, 1u);
        is_last = (prev == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last)
        return;
    __threadfence();
    double acc = 0.0;
    for (uint32_t b = tid; b < gridDim.x; b += PR_BLOCK)
        acc += ld_agent(&blk_err[b]);
    __syncthreads(); // red[] is reused
    const double sweep = block_sum<double, PR_WAVES>(acc, red);
    if (tid == 0) {
        *err_out = sweep;
        st_agent(ticket, 0u);
    }
}

// ---- synchronous sweep with the reference's per-row summation order --------------------------------
// One lane per row adds out_scores[v] left to right in CSR order with one f32 add per edge — the exact
// rounding of page_rank.rs:143-146.  Hub rows make it slow; it exists to show that the only
// difference between the fast engines and the reference at the fixed point is this summation order.
__global__ __launch_bounds__(PR_BLOCK) void pr_reforder_kernel(const uint32_t *__restrict__ off,
                                                               const uint32_t *__restrict__ tgt,
                                                               const float *__restrict__ x_in,
                                                               const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                                               float *__restrict__ x_out, double *__restrict__ blk_err,
                                                               uint32_t n, float base, float damping)
{
    __shared__ double red[PR_WAVES];
    const uint32_t stride = gridDim.x * blockDim.x;
    double err = 0.0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        float sum = 0.0f;
        for (uint32_t i = off[r]; i < off[r + 1]; ++i)
            sum = __fadd_rn(sum, x_in[tgt[i]]);
        err += pr_finalize(r, sum, base, damping, outdeg, scores, x_out);
    }
    const double total = block_sum<double, PR_WAVES>(err, red);
    if (threadIdx.x == 0)
        blk_err[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void pr_sum_partials_kernel(const double *__restrict__ part, uint32_t count,
                                                               double *__restrict__ err_out)
{
    __shared__ double red[1024 / kWave];
    double acc = 0.0;
    for (uint32_t b = threadIdx.x; b < count; b += 1024)
        acc += part[b];
    const double total = block_sum<double, 1024 / kWave>(acc, red);
    if (threadIdx.x == 0)
        *err_out = total;
}

// ---- the reference's exact sequential order on one wavefront ----------------------------------
// X_IN_LDS: out_scores live in LDS (n <= 16384 -> 64 KiB); otherwise in global memory, accessed
// with L1-bypassing loads/stores so every lane sees lane 0's in-place update.
template <bool X_IN_LDS>
__global__ __launch_bounds__(kWave) void pr_seq_kernel(const uint32_t *__restrict__ off,
                                                       const uint32_t *__restrict__ tgt,
                                                       const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                                       float *xg, uint32_t n, float init, float base, float damping,
                                                       uint64_t max_iterations, double tolerance,
                                                       uint64_t *__restrict__ iterations_out,
                                                       double *__restrict__ error_out)
{
    extern __shared__ float xs[];
    const uint32_t lane = threadIdx.x;
    for (uint32_t u = lane; u < n; u += kWave) {
        scores[u] = init;
        const float o = __fdiv_rn(init, (float)outdeg[u]);
        if (X_IN_LDS)
            xs[u] = o;
        else
            st_agent(&xg[u], o);
    }
    __syncthreads();
    uint64_t iter = 0;
    double err = 0.0;
    for (;;) {
        err = 0.0;
        for (uint32_t u = 0; u < n; ++u) {
            const uint32_t s = off[u], e = off[u + 1];
            float sum = 0.0f;
            for (uint32_t c = s; c < e; c += kWave) {
                const uint32_t idx = c + lane;
                float v = 0.0f;
                if (idx < e) {
                    const uint32_t w = tgt[idx];
                    v = X_IN_LDS ? xs[w] : ld_agent(&xg[w]);
                }
                const uint32_t cnt = (e - c) < (uint32_t)kWave ? (e - c) : (uint32_t)kWave;
                for (uint32_t i = 0; i < cnt; ++i) // CSR order, one f32 add per edge (page_rank.rs:143-146)
                    sum = __fadd_rn(sum, __shfl(v, (int)i, kWave));
            }
            const float old = scores[u];
            const float nw = pr_new_score(base, damping, sum);
            err += fabs((double)__fsub_rn(nw, old));
            __syncthreads(); // every lane has read scores[u] / x before lane 0 overwrites them
            if (lane == 0) {
                scores[u] = nw;
                const float o = __fdiv_rn(nw, (float)outdeg[u]);
                if (X_IN_LDS)
                    xs[u] = o;
                else
                    st_agent(&xg[u], o);
            }
            __syncthreads();
        }
        iter += 1;
        if (err < tolerance || iter == max_iterations)
            break;
    }
    if (lane == 0) {
        *iterations_out = iter;
        *error_out = err;
    }
}

} // namespace

// Does some row have at least GM_PB_HUB_DEG (default 4096) in-edges?  Such rows need the propagation-blocking
// engine, which sums them in the reference's left-to-right f32 order (pagerank_pb.hip): the pull tiles reduce a
// long row as a tree, and on long rows the two differ by more than the 1e-5 the results must agree to
// (measured 1.2e-5 at RMAT scale 18).  Looked at once per handle and threshold.
static int pr_has_long_rows(const gm_csr *csr, bool whole, bool *out)
{
    const char *v = getenv("GM_PB_HUB_DEG");
    const long thr = v && *v && atol(v) > 0 ? atol(v) : (v && *v ? 0 : 4096);
    const long leaf_t = thr > 0 ? (long)gm::hub_leaves_threshold() : 0; // (GM_PB_HUB_DEG=0: every row exactly rounded, no rule either)
    const bool have_flags = csr->source_flags.p && csr->source_flags_len;
    const long long key = ((long long)thr << 21) | ((long long)(leaf_t & 0xFFFFF) << 1) | (have_flags ? 1 : 0);
    const long long cached = csr->long_rows.load();
    int state = cached >= 0 && (cached >> 1) == key ? (int)(cached & 1) : -1; // an answer for other thresholds does not count
    if (state < 0) {
        state = 0;
        if (thr > 0 && csr->n && csr->m >= (uint64_t)(leaf_t && leaf_t < thr ? leaf_t : thr)) {
            gm::DevBuf flag;
            GM_TRY(flag.alloc(4));
            GM_HIP(hipMemset(flag.p, 0, 4));
            unsigned grid = gm::div_up(csr->n, 256);
            // the cheap question first (a row of thr entries?); the lists of the rows between leaf_t and thr entries are walked only if not
            uint32_t h = 0;
            for (int pass = 0; pass < 2 && !h; ++pass) {
                const uint32_t lt = pass == 0 ? 0u : (uint32_t)(leaf_t < thr ? leaf_t : 0);
                if (pass == 1 && !lt)
                    break;
                hipLaunchKernelGGL(pr_long_row_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, 0, csr->offsets, csr->targets,
                                   (uint32_t)csr->n, (uint32_t)thr, lt, have_flags ? csr->source_flags.as<uint8_t>() : (const uint8_t *)nullptr,
                                   csr->source_flags_len, whole ? 1 : 0, flag.as<uint32_t>());
                GM_HIP(hipGetLastError());
                GM_HIP(hipMemcpy(&h, flag.p, 4, hipMemcpyDeviceToHost));
            }
            state = h ? 1 : 0;
        }
        csr->long_rows.store((key << 1) | (long long)state);
    }
    *out = state == 1;
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
GM_API int gm_pr_create_with(const gm_csr *csr, uint64_t n_global, uint64_t row_begin, uint64_t x_len,
                             uint64_t d_out_degree_local, float damping_factor, int engine, gm_pr **out)
{
    GM_CHECK(csr && out, GM_ERR_INVALID, "gm_pr_create: null argument");
    GM_CHECK(n_global > 0 && row_begin + csr->n <= n_global, GM_ERR_INVALID,
             "gm_pr_create: rows [%llu, %llu) outside a graph of %llu nodes", (unsigned long long)row_begin,
             (unsigned long long)(row_begin + csr->n), (unsigned long long)n_global);
    GM_CHECK(d_out_degree_local != 0 || csr->n == 0, GM_ERR_INVALID, "gm_pr_create: out-degree pointer is null");
    GM_CHECK(csr->n + csr->m < (1ull << 32), GM_ERR_RANGE, "gm_pr_create: n + m = %llu does not fit 32-bit tile positions",
             (unsigned long long)(csr->n + csr->m));
    // x_len may be smaller than n_global: a compacted exchange buffer only holds nodes with out-edges
    GM_CHECK(x_len >= 1 && x_len < (1ull << 32), GM_ERR_RANGE, "gm_pr_create: x_len %llu out of range",
             (unsigned long long)x_len);
    GM_CHECK(engine >= GM_PR_ENGINE_AUTO && engine <= GM_PR_ENGINE_REFORDER, GM_ERR_INVALID, "gm_pr_create: unknown engine %d", engine);
    gm::DeviceGuard guard(csr->device);
    if (engine == GM_PR_ENGINE_AUTO) { // below ~16M edges the gathered vector is cache-resident: the pull tiles win
        bool long_rows = false;         // ... unless a long row needs the reference's summation order
        if (csr->m < (1ull << 24))
            GM_TRY(pr_has_long_rows(csr, x_len == csr->n, &long_rows));
        engine = (csr->m >= (1ull << 24) || long_rows) ? GM_PR_ENGINE_PB : GM_PR_ENGINE_PULL;
    }
    gm_pr *pr = new (std::nothrow) gm_pr();
    GM_CHECK(pr, GM_ERR_NOMEM, "gm_pr_create: out of host memory");
    pr->csr = csr;
    pr->n_global = n_global;
    pr->row_begin = row_begin;
    pr->x_len = x_len;
    pr->engine = engine;
    pr->n_local = (uint32_t)csr->n;
    pr->m = (uint32_t)csr->m;
    pr->outdeg = reinterpret_cast<const uint32_t *>(d_out_degree_local);
    pr->damping = damping_factor;
    // page_rank.rs:70-71: init_score = 1/n, base_score = (1 - damping)/n, in f32
    pr->init = 1.0f / (float)n_global;
    pr->base = (1.0f - damping_factor) / (float)n_global;
    if (engine == GM_PR_ENGINE_PB) {
        // GM_PB_EARLY_VALS=1 (measurement): reserve the value stream before the plan is built — its pages decide 15 % of
        // the sweep time; does it matter whether they are drawn before or after the build's 30 GB of churn?
        gm::DevBuf early;
        if (getenv("GM_PB_EARLY_VALS") && atoi(getenv("GM_PB_EARLY_VALS")) == 1)
            (void)early.alloc((size_t)csr->m * 4 + ((size_t)csr->m >> 3) + (64u << 20));
        int rc = gm::pb_plan_get(csr, x_len, &pr->pb_keep);
        pr->pb = pr->pb_keep.get();
        if (rc == GM_OK)
            rc = gm::pb_scratch_create(pr->pb, &pr->pb_scratch, &early);
        if (rc != GM_OK) {
            delete pr;
            return rc;
        }
        pr->T = (uint32_t)gm::pb_work_items(pr->pb);
        *out = pr;
        return GM_OK;
    }
    if (engine == GM_PR_ENGINE_REFORDER) {
        pr->G = gm::div_up(csr->n ? csr->n : 1, PR_BLOCK);
        if (pr->G > 8192)
            pr->G = 8192;
        pr->T = pr->G;
        const int rc = pr->blk_err.alloc((size_t)pr->G * 8);
        if (rc != GM_OK) {
            delete pr;
            return rc;
        }
        *out = pr;
        return GM_OK;
    }
    const uint64_t items = csr->n + csr->m;
    pr->T = (uint32_t)((items + PR_W - 1) / PR_W);
    if (pr->T == 0)
        pr->T = 1;
    pr->G = gm::div_up(pr->T, PR_BLOCK);
    int rc = GM_OK;
    if ((rc = pr->tile_row.alloc(((size_t)pr->T + 1) * 4)) || (rc = pr->head.alloc((size_t)pr->T * 4)) ||
        (rc = pr->tail.alloc((size_t)pr->T * 4)) || (rc = pr->tile_err.alloc((size_t)pr->T * 8)) ||
        (rc = pr->blk_err.alloc((size_t)pr->G * 8)) || (rc = pr->ticket.alloc(16))) {
        delete pr;
        return rc;
    }
    hipError_t e = hipMemset(pr->ticket.p, 0, 16);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pr_tile_rows_kernel, dim3(gm::div_up((uint64_t)pr->T + 1, 256)), dim3(256), 0, 0,
                           csr->offsets, pr->n_local, pr->T, pr->tile_row.as<uint32_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        gm::set_error("gm_pr_create: %s", hipGetErrorString(e));
        delete pr;
        return GM_ERR_HIP;
    }
    *out = pr;
    return GM_OK;
}

GM_API int gm_pr_create(const gm_csr *csr, uint64_t n_global, uint64_t row_begin, uint64_t d_out_degree_local,
                        float damping_factor, gm_pr **out)
{
    return gm_pr_create_with(csr, n_global, row_begin, n_global, d_out_degree_local, damping_factor, GM_PR_ENGINE_AUTO,
                             out);
}

GM_API void gm_pr_destroy(gm_pr *pr) { delete pr; }

GM_API uint64_t gm_pr_algorithmic_bytes(const gm_pr *pr)
{
    return pr ? 8ull * pr->m + 20ull * pr->n_local + 4ull : 0;
}

GM_API uint64_t gm_pr_tile_count(const gm_pr *pr) { return pr ? pr->T : 0; }
GM_API int gm_pr_engine(const gm_pr *pr) { return pr ? pr->engine : 0; }

GM_API int gm_pr_init(gm_pr *pr, uint64_t d_scores_local, uint64_t d_x_local, void *stream)
{
    GM_CHECK(pr, GM_ERR_INVALID, "gm_pr_init: null engine");
    if (pr->n_local == 0)
        return GM_OK;
    gm::DeviceGuard guard(pr->csr->device);
    unsigned grid = gm::div_up(pr->n_local, 256);
    if (grid > 256 * 8)
        grid = 256 * 8;
    hipLaunchKernelGGL(pr_init_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pr->n_local, pr->init,
                       pr->outdeg, reinterpret_cast<float *>(d_scores_local), reinterpret_cast<float *>(d_x_local));
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_pr_sweep_tiles(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                             void *stream)
{
    GM_CHECK(pr, GM_ERR_INVALID, "gm_pr_sweep_tiles: null engine");
    gm::DeviceGuard guard(pr->csr->device);
    if (pr->engine == GM_PR_ENGINE_REFORDER) {
        hipLaunchKernelGGL(pr_reforder_kernel, dim3(pr->G), dim3(PR_BLOCK), 0, (hipStream_t)stream, pr->csr->offsets,
                           pr->csr->targets, reinterpret_cast<const float *>(d_x_in_global), pr->outdeg,
                           reinterpret_cast<float *>(d_scores_local), reinterpret_cast<float *>(d_x_out_local),
                           pr->blk_err.as<double>(), pr->n_local, pr->base, pr->damping);
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
    if (pr->engine == GM_PR_ENGINE_PB)
        return gm::pb_sweep_main(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global),
                                 reinterpret_cast<float *>(d_x_out_local), reinterpret_cast<float *>(d_scores_local),
                                 pr->outdeg, pr->base, pr->damping, (hipStream_t)stream);
    hipLaunchKernelGGL(pr_tile_kernel, dim3(pr->T), dim3(PR_BLOCK), 0, (hipStream_t)stream, pr->csr->offsets,
                       pr->csr->targets, pr->tile_row.as<uint32_t>(), reinterpret_cast<const float *>(d_x_in_global),
                       pr->outdeg, reinterpret_cast<float *>(d_scores_local), reinterpret_cast<float *>(d_x_out_local),
                       pr->head.as<float>(), pr->tail.as<float>(), pr->tile_err.as<double>(), pr->m, pr->base,
                       pr->damping);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_pr_sweep_fixup(gm_pr *pr, uint64_t d_x_out_local, uint64_t d_scores_local, uint64_t d_error_out,
                             void *stream)
{
    GM_CHECK(pr && d_error_out, GM_ERR_INVALID, "gm_pr_sweep_fixup: null argument");
    gm::DeviceGuard guard(pr->csr->device);
    if (pr->engine == GM_PR_ENGINE_REFORDER) {
        hipLaunchKernelGGL(pr_sum_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pr->blk_err.as<double>(),
                           pr->G, reinterpret_cast<double *>(d_error_out));
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
    if (pr->engine == GM_PR_ENGINE_PB)
        return gm::pb_sweep_error(pr->pb, pr->pb_scratch, reinterpret_cast<double *>(d_error_out), (hipStream_t)stream);
    hipLaunchKernelGGL(pr_fixup_kernel, dim3(pr->G), dim3(PR_BLOCK), 0, (hipStream_t)stream, pr->csr->offsets,
                       pr->tile_row.as<uint32_t>(), pr->outdeg, reinterpret_cast<float *>(d_scores_local),
                       reinterpret_cast<float *>(d_x_out_local), pr->head.as<float>(), pr->tail.as<float>(),
                       pr->tile_err.as<double>(), pr->blk_err.as<double>(), pr->ticket.as<uint32_t>(),
                       reinterpret_cast<double *>(d_error_out), pr->T, pr->m, pr->base, pr->damping);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_pr_part_geometry(const gm_pr *pr, uint64_t *rows_per_bin_out, uint64_t *source_tile_out)
{
    GM_CHECK(pr && rows_per_bin_out && source_tile_out, GM_ERR_INVALID, "gm_pr_part_geometry: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_part_geometry: not a propagation-blocking engine");
    *rows_per_bin_out = gm::pb_rows_per_bin(pr->pb);
    *source_tile_out = gm::pb_source_tile(pr->pb);
    return GM_OK;
}

GM_API int gm_pr_set_parts(gm_pr *pr, const uint64_t *row_splits, uint64_t n_parts)
{
    GM_CHECK(pr && row_splits, GM_ERR_INVALID, "gm_pr_set_parts: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_set_parts: not a propagation-blocking engine");
    GM_CHECK(n_parts >= 1 && n_parts <= 64, GM_ERR_INVALID, "gm_pr_set_parts: %llu parts", (unsigned long long)n_parts);
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_set_parts(pr->pb, pr->pb_scratch, row_splits, (uint32_t)n_parts);
}

GM_API int gm_pr_sweep_bin(gm_pr *pr, uint64_t d_x_in_global, uint64_t x_lo, uint64_t x_hi, void *stream)
{
    GM_CHECK(pr && d_x_in_global, GM_ERR_INVALID, "gm_pr_sweep_bin: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_sweep_bin: not a propagation-blocking engine");
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_sweep_bin_range(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global), x_lo, x_hi,
                                  (hipStream_t)stream);
}

GM_API int gm_pr_set_bin_regions(gm_pr *pr, const uint64_t *x_lo, const uint64_t *x_hi, const uint32_t *region, uint64_t count,
                                 uint32_t n_regions)
{
    GM_CHECK(pr && (count == 0 || (x_lo && x_hi && region)), GM_ERR_INVALID, "gm_pr_set_bin_regions: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_set_bin_regions: not a propagation-blocking engine");
    GM_CHECK(n_regions >= 1 && n_regions <= 64 && count <= 65536, GM_ERR_INVALID, "gm_pr_set_bin_regions: %u regions, %llu ranges",
             n_regions, (unsigned long long)count);
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_set_regions(pr->pb, pr->pb_scratch, x_lo, x_hi, region, (uint32_t)count, n_regions);
}

GM_API int gm_pr_sweep_bin_region(gm_pr *pr, uint64_t d_x_in_global, uint32_t region, void *stream)
{
    GM_CHECK(pr && d_x_in_global, GM_ERR_INVALID, "gm_pr_sweep_bin_region: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_sweep_bin_region: not a propagation-blocking engine");
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_sweep_bin_region(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global), region, (hipStream_t)stream);
}

GM_API int gm_pr_plan_info(const gm_pr *pr, uint64_t *info, uint32_t count)
{
    GM_CHECK(pr && info, GM_ERR_INVALID, "gm_pr_plan_info: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_plan_info: not a propagation-blocking engine");
    gm::pb_plan_info(pr->pb, pr->pb_scratch, info, count);
    return GM_OK;
}

GM_API int gm_pr_sweep_hot(gm_pr *pr, uint64_t d_x_in_global, void *stream)
{
    GM_CHECK(pr && d_x_in_global, GM_ERR_INVALID, "gm_pr_sweep_hot: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_sweep_hot: not a propagation-blocking engine");
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_sweep_hot(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global), (hipStream_t)stream);
}

GM_API int gm_pr_sweep_accum(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                             uint64_t part, int stage_hot, void *stream)
{
    GM_CHECK(pr && d_x_in_global, GM_ERR_INVALID, "gm_pr_sweep_accum: null argument");
    GM_CHECK(pr->engine == GM_PR_ENGINE_PB, GM_ERR_UNSUPPORTED, "gm_pr_sweep_accum: not a propagation-blocking engine");
    GM_CHECK(part < 64, GM_ERR_INVALID, "gm_pr_sweep_accum: part %llu", (unsigned long long)part);
    gm::DeviceGuard guard(pr->csr->device);
    return gm::pb_sweep_accum_part(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global),
                                   reinterpret_cast<float *>(d_x_out_local), reinterpret_cast<float *>(d_scores_local),
                                   pr->outdeg, pr->base, pr->damping, (uint32_t)part, stage_hot, (hipStream_t)stream);
}

GM_API int gm_pr_sweep(gm_pr *pr, uint64_t d_x_in_global, uint64_t d_x_out_local, uint64_t d_scores_local,
                       uint64_t d_error_out, void *stream)
{
    GM_CHECK(pr && d_error_out, GM_ERR_INVALID, "gm_pr_sweep: null argument");
    if (pr->engine == GM_PR_ENGINE_PB) { // the error comes out of the sweep's own launches where it can (pb_err_fold)
        gm::DeviceGuard guard(pr->csr->device);
        bool folded = false;
        GM_TRY(gm::pb_sweep_main(pr->pb, pr->pb_scratch, reinterpret_cast<const float *>(d_x_in_global),
                                 reinterpret_cast<float *>(d_x_out_local), reinterpret_cast<float *>(d_scores_local), pr->outdeg,
                                 pr->base, pr->damping, (hipStream_t)stream, reinterpret_cast<double *>(d_error_out), &folded));
        if (folded)
            return GM_OK;
        return gm::pb_sweep_error(pr->pb, pr->pb_scratch, reinterpret_cast<double *>(d_error_out), (hipStream_t)stream);
    }
    GM_TRY(gm_pr_sweep_tiles(pr, d_x_in_global, d_x_out_local, d_scores_local, stream));
    return gm_pr_sweep_fixup(pr, d_x_out_local, d_scores_local, d_error_out, stream);
}

// ------------------------------------------------------------------------------------------------
// gm_page_rank: the whole `page_rank()` call on one GPU with host result buffers.
// ------------------------------------------------------------------------------------------------
namespace {

// out-degrees on the device: from the out-CSR's offsets when the caller has one (no PCIe traffic), else
// the caller's host array, else counted from the in-CSR's targets
int pr_out_degrees(const gm_csr *csr, const uint32_t *host_outdeg, const gm_csr *out_csr, gm::DevBuf &buf, hipStream_t st)
{
    if (buf.bytes < (size_t)csr->n * 4 || !buf.p)
        GM_TRY(buf.alloc((size_t)csr->n * 4));
    if (out_csr) {
        unsigned grid = gm::div_up(csr->n, 256);
        if (grid > 256 * 16)
            grid = 256 * 16;
        if (csr->n)
            hipLaunchKernelGGL(pr_degrees_from_offsets_kernel, dim3(grid), dim3(256), 0, st, out_csr->offsets, (uint32_t)csr->n,
                               buf.as<uint32_t>());
        GM_HIP(hipGetLastError());
    } else if (host_outdeg) {
        GM_HIP(hipMemcpyAsync(buf.p, host_outdeg, (size_t)csr->n * 4, hipMemcpyHostToDevice, st));
    } else {
        GM_HIP(hipMemsetAsync(buf.p, 0, (size_t)csr->n * 4, st));
        if (csr->m) {
            unsigned grid = gm::div_up(csr->m, 256);
            if (grid > 256 * 16)
                grid = 256 * 16;
            hipLaunchKernelGGL(pr_count_outdeg_kernel, dim3(grid), dim3(256), 0, st, csr->targets, csr->m,
                               buf.as<uint32_t>());
            GM_HIP(hipGetLastError());
        }
    }
    return GM_OK;
}

} // namespace

// Everything a gm_page_rank call allocates: its stream, the out-degree / score / x vectors, the read-back words and
// the engine (with the propagation-blocking scratch: 3.6 GB at scale 26).  A call takes the set parked in the in-CSR's
// handle, grows what is too small, rebuilds the engine if the call wants another one, and parks the set again:
// a dozen hipFree + hipMalloc pairs were ~2 ms of every call (a third of a 20-sweep call at scale 22), and the
// reference's app calls page_rank(&graph, config) in a loop on one graph (crates/app/src/app.rs:124-153).
struct gm::PrCallState {
    hipStream_t stream = nullptr;
    gm::DevBuf outdeg, scores, x0, x1, dres;
    gm::PinnedBuf hres;
    gm_pr *engine = nullptr;
    std::vector<uint64_t> gs_splits; // row blocks of the block-Gauss-Seidel sweeps the parked engine is set up for
    ~PrCallState()
    {
        delete engine;
        if (stream)
            (void)hipStreamDestroy(stream);
    }
};

static int page_rank_impl(const gm_csr *in_csr, const uint32_t *out_degree, const gm_csr *out_csr, uint64_t max_iterations,
                          double tolerance, float damping_factor, int mode, float *scores_out, uint64_t *iterations_out,
                          double *error_out)
{
    GM_CHECK(in_csr && iterations_out && error_out, GM_ERR_INVALID, "gm_page_rank: null argument");
    GM_CHECK(!out_csr || (out_csr->n == in_csr->n && out_csr->m == in_csr->m && out_csr->device == in_csr->device),
             GM_ERR_INVALID, "gm_page_rank_directed: the two CSRs are not the out- and in-lists of one graph on one device");
    GM_CHECK(mode >= GM_PR_AUTO && mode <= GM_PR_BLOCK_GS, GM_ERR_INVALID, "gm_page_rank: unknown mode %d", mode);
    // page_rank.rs:105-109: the loop only ends on error < tolerance or iteration == max_iterations
    GM_CHECK(max_iterations != 0 || tolerance > 0.0, GM_ERR_INVALID,
             "gm_page_rank: max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)");
    const uint64_t n = in_csr->n;
    if (n == 0) { // one empty sweep: error 0.0
        GM_CHECK(0.0 < tolerance || max_iterations == 1, GM_ERR_INVALID,
                 "gm_page_rank: empty graph with tolerance <= 0 never terminates (reference: infinite loop)");
        *iterations_out = 1;
        *error_out = 0.0;
        return GM_OK;
    }
    GM_CHECK(scores_out, GM_ERR_INVALID, "gm_page_rank: scores_out is null");
    const bool auto_mode = mode == GM_PR_AUTO;
    if (mode == GM_PR_AUTO)
        mode = n <= 16384 ? GM_PR_SEQUENTIAL : GM_PR_JACOBI;

    gm::DeviceGuard guard(in_csr->device);
    // GM_PB_NOCACHE (measurement tools that switch plan knobs between calls): nothing is carried over
    const bool carry = getenv("GM_PB_NOCACHE") == nullptr || atoi(getenv("GM_PB_NOCACHE")) == 0;
    std::shared_ptr<gm::PrCallState> cs;
    if (carry) {
        std::lock_guard<std::mutex> lock(in_csr->cache_mu);
        cs = std::move(in_csr->pr_call);
        in_csr->pr_call.reset();
    }
    struct Park { // back into the handle on every way out
        const gm_csr *g;
        std::shared_ptr<gm::PrCallState> &cs;
        bool carry;
        ~Park()
        {
            if (!carry || !cs)
                return;
            std::lock_guard<std::mutex> lock(g->cache_mu);
            if (!g->pr_call)
                g->pr_call = std::move(cs);
        }
    } park{in_csr, cs, carry};
    if (!cs)
        cs = std::make_shared<gm::PrCallState>();
    if (!cs->stream)
        GM_HIP(hipStreamCreateWithFlags(&cs->stream, hipStreamNonBlocking));
    hipStream_t st = cs->stream;

    gm::DevBuf &outdeg = cs->outdeg, &scores = cs->scores, &x0 = cs->x0, &x1 = cs->x1, &dres = cs->dres;
    gm::PinnedBuf &hres = cs->hres;
    GM_TRY(pr_out_degrees(in_csr, out_degree, out_csr, outdeg, st));
    if (scores.bytes < n * 4)
        GM_TRY(scores.alloc(n * 4));
    if (!dres.p)
        GM_TRY(dres.alloc(16));
    if (!hres.p)
        GM_TRY(hres.alloc(16));

    const float init = 1.0f / (float)n;
    const float base = (1.0f - damping_factor) / (float)n;

    if (mode == GM_PR_SEQUENTIAL) {
        const bool lds = n <= 16384;
        if (!lds && x0.bytes < n * 4)
            GM_TRY(x0.alloc(n * 4));
        uint64_t *d_iter = dres.as<uint64_t>();
        double *d_err = reinterpret_cast<double *>(dres.as<char>() + 8);
        if (lds)
            GM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pr_seq_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4));
        if (lds)
            hipLaunchKernelGGL(pr_seq_kernel<true>, dim3(1), dim3(kWave), n * 4, st, in_csr->offsets,
                               in_csr->targets, outdeg.as<uint32_t>(), scores.as<float>(), (float *)nullptr,
                               (uint32_t)n, init, base, damping_factor, max_iterations, tolerance, d_iter, d_err);
        else
            hipLaunchKernelGGL(pr_seq_kernel<false>, dim3(1), dim3(kWave), 0, st, in_csr->offsets, in_csr->targets,
                               outdeg.as<uint32_t>(), scores.as<float>(), x0.as<float>(), (uint32_t)n, init, base,
                               damping_factor, max_iterations, tolerance, d_iter, d_err);
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(hres.p, dres.p, 16, hipMemcpyDeviceToHost, st));
        GM_HIP(hipMemcpyAsync(scores_out, scores.p, n * 4, hipMemcpyDeviceToHost, st));
        GM_HIP(hipStreamSynchronize(st));
        *iterations_out = hres.as<uint64_t>()[0];
        *error_out = hres.as<double>()[1];
        return GM_OK;
    }

    // synchronous sweeps
    if (x0.bytes < n * 4)
        GM_TRY(x0.alloc(n * 4));
    if (x1.bytes < n * 4)
        GM_TRY(x1.alloc(n * 4));
    struct { gm_pr *p; } ph{nullptr};
    int engine = mode == GM_PR_JACOBI_PULL       ? GM_PR_ENGINE_PULL
                 : mode == GM_PR_JACOBI_PB       ? GM_PR_ENGINE_PB
                 : mode == GM_PR_BLOCK_GS        ? GM_PR_ENGINE_PB
                 : mode == GM_PR_JACOBI_REFORDER ? GM_PR_ENGINE_REFORDER
                                                 : GM_PR_ENGINE_AUTO;
    if (engine == GM_PR_ENGINE_AUTO) {
        // One-shot economics: building the propagation-blocking plan costs ~27 ms at 2^26 edges and ~120 ms
        // at 2^30, a pull sweep ~m / 100 G/s (far less beyond 2^28 edges), a PB sweep ~m / 350 G/s.  So: use the
        // plan if this handle already has one; build it straight away where ~20 sweeps repay it (>= 2^28 edges); otherwise
        // run this call on the pull tiles and build the plan on the second call of the same graph (the
        // reference's app runs 5 warm-ups + N timed runs on one graph, crates/app/src/app.rs:124-153).
        bool long_rows = false; // rows that must be summed in the reference's order: only the PB engine does that
        if (in_csr->m < (1ull << 28))
            GM_TRY(pr_has_long_rows(in_csr, true, &long_rows));
        std::lock_guard<std::mutex> lock(in_csr->cache_mu);
        const uint64_t calls = ++in_csr->page_rank_calls;
        const bool cached = in_csr->pb_plans.count(n) != 0;
        const bool big = in_csr->m >= (1ull << 28), mid = in_csr->m >= (1ull << 24);
        engine = (cached || big || long_rows || (mid && calls >= 2)) ? GM_PR_ENGINE_PB : GM_PR_ENGINE_PULL;
    }
    // the parked engine serves this call if it is the same kind over the same vectors
    if (cs->engine && (cs->engine->engine != engine || cs->engine->damping != damping_factor || cs->engine->n_global != n ||
                       cs->engine->outdeg != outdeg.as<uint32_t>())) {
        delete cs->engine;
        cs->engine = nullptr;
    }
    if (!cs->engine)
        cs->gs_splits.clear(); // (a new engine has no row blocks yet)
    if (!cs->engine)
        GM_TRY(gm_pr_create_with(in_csr, n, 0, n, (uint64_t)outdeg.p, damping_factor, engine, &cs->engine));
    ph.p = cs->engine;
    // Block-Gauss-Seidel sweeps (GM_PR_BLOCK_GS; what GM_PR_AUTO runs on the propagation-blocking engine unless
    // GM_PR_BLOCK_GS=0).  The reference updates out_scores IN PLACE while it sweeps the nodes in ascending chunks
    // (page_rank.rs:142-160): a row sees this sweep's values of the rows before it.  Synchronous sweeps need about twice as
    // many iterations for the same error, so with PageRankConfig::default() (20 / 1e-4, page_rank.rs:17-56) they run out of
    // iterations where the reference stops on its tolerance.  Here the rows are cut into K blocks of about equal in-edges
    // (whole source tiles); per sweep and block j, in order: accumulate and finish the rows of block j (from the value stream
    // as it stands: this sweep's out_scores of the blocks before j, the last sweep's of the others), then propagate block j's new
    // out_scores into the value stream.  One vector, updated in place; a hub row is summed with the block its row lies in
    // (pb_set_parts(.., hub_by_part): with all of them summed beside block 0 from the last sweep's values, the rows that carry
    // most of the error converged at the synchronous rate — 20 iterations at scale 22 instead of 16).  Every row's equation is
    // the one of page_rank.rs:143-159, only WHICH sweep's value a term carries differs — as it does between two runs of the
    // reference itself.  Deterministic; same fixed point.  K = 8 blocks (GM_PR_BLOCK_GS=K): every block costs a hub launch whose
    // lane walks are a fixed ~0.1 ms (scale 22) to ~0.35 ms (scale 26) of latency, and every halving of K one iteration more —
    // measured with PageRankConfig::default() (tools/runs/r06_call15.sh): scale 22 K = 16 / 8 / 4: 15 / 16 / 17 iterations in
    // 31 / 17 / 11 ms (reference: 14 iterations; synchronous sweeps: out of their 20 in 4.7 ms); scale 26: 14 / 14 / 15
    // iterations in 118 / 73 / 54 ms (synchronous: 20 in 66 ms).
    std::vector<uint64_t> gs_splits;
    {
        const char *gs_env_s = getenv("GM_PR_BLOCK_GS");
        const int gs_env = gs_env_s ? atoi(gs_env_s) : -1;
        const bool want = ph.p->engine == GM_PR_ENGINE_PB && (mode == GM_PR_BLOCK_GS || (auto_mode && gs_env != 0));
        uint64_t rows_per_bin = 0, tile = 0;
        if (want)
            GM_TRY(gm_pr_part_geometry(ph.p, &rows_per_bin, &tile));
        const uint64_t tiles = tile ? (n + tile - 1) / tile : 0;
        uint64_t K = gs_env > 1 ? (uint64_t)gs_env : 8;
        if (K > 64)
            K = 64;
        if (K > tiles)
            K = tiles;
        if (want && K >= 2) {
            if (cs->gs_splits.size() == K + 1 && cs->gs_splits.back() == n) {
                gs_splits = cs->gs_splits; // the parked engine is set up for these blocks
            } else {
                // block j ends at the first tile boundary with at least j / K of the in-edges in front of it
                gs_splits.assign(K + 1, 0);
                gs_splits[K] = n;
                for (uint64_t j = 1; j < K; ++j) {
                    const uint64_t want_edges = in_csr->m / K * j;
                    uint64_t lo = gs_splits[j - 1] / tile, hi = tiles; // answer in (lo, hi]
                    while (hi - lo > 1) {
                        const uint64_t mid = (lo + hi) / 2;
                        uint32_t off = 0;
                        GM_HIP(hipMemcpy(&off, in_csr->offsets + mid * tile, 4, hipMemcpyDeviceToHost));
                        if (off >= want_edges)
                            hi = mid;
                        else
                            lo = mid;
                    }
                    gs_splits[j] = hi * tile < n ? hi * tile : n;
                }
                // (GM_PR_GS_HUBS=0, measurements: every hub row beside block 0, from the last sweep's values — round 6's first version)
                const bool hubs_by_block = !(gm::measure_env("GM_PR_GS_HUBS") && atoi(gm::measure_env("GM_PR_GS_HUBS")) == 0);
                GM_TRY(gm::pb_set_parts(ph.p->pb, ph.p->pb_scratch, gs_splits.data(), (uint32_t)K, hubs_by_block));
                std::vector<uint64_t> lo(gs_splits.begin(), gs_splits.end() - 1), hi(gs_splits.begin() + 1, gs_splits.end());
                std::vector<uint32_t> reg(K);
                for (uint64_t j = 0; j < K; ++j)
                    reg[j] = (uint32_t)j;
                GM_TRY(gm_pr_set_bin_regions(ph.p, lo.data(), hi.data(), reg.data(), K, (uint32_t)K));
                cs->gs_splits = gs_splits;
            }
        }
    }
    GM_TRY(gm_pr_init(ph.p, (uint64_t)scores.p, (uint64_t)x0.p, st));
    uint64_t iter = 0;
    double err = 0.0;
    float *xin = x0.as<float>(), *xout = x1.as<float>();
    const bool can_stop_early = tolerance > 0.0; // error >= 0 always
    const uint64_t gs_blocks = gs_splits.empty() ? 0 : gs_splits.size() - 1;
    if (gs_blocks) {
        xout = xin; // in place
        GM_TRY(gm_pr_sweep_bin(ph.p, (uint64_t)xin, 0, n, st)); // the initial out_scores of every block (page_rank.rs:70-81)
    }
    for (;;) {
        gm::PhaseTimer timer(st);
        if (gs_blocks) {
            for (uint64_t j = 0; j < gs_blocks; ++j) {
                GM_TRY(gm_pr_sweep_accum(ph.p, (uint64_t)xin, (uint64_t)xin, (uint64_t)scores.p, j, 1, st));
                GM_TRY(gm_pr_sweep_bin_region(ph.p, (uint64_t)xin, (uint32_t)j, st));
            }
            GM_TRY(gm_pr_sweep_fixup(ph.p, (uint64_t)xin, (uint64_t)scores.p, (uint64_t)dres.p, st));
        } else
            GM_TRY(gm_pr_sweep(ph.p, (uint64_t)xin, (uint64_t)xout, (uint64_t)scores.p, (uint64_t)dres.p, st));
        if (gm::log_enabled()) { // page_rank.rs:95-100: "Finished iteration {} with an error of {:.6} in {:?}"
            double e = 0.0;
            (void)hipMemcpyAsync(&e, dres.p, 8, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            timer.done("Finished iteration %llu with an error of %.6f;", (unsigned long long)iter, e);
        }
        iter += 1;
        float *tmp = xin;
        xin = xout;
        xout = tmp;
        const bool last = iter == max_iterations;
        if (can_stop_early || last) {
            GM_HIP(hipMemcpyAsync(hres.p, dres.p, 8, hipMemcpyDeviceToHost, st));
            GM_HIP(hipStreamSynchronize(st));
            err = hres.as<double>()[0];
            if (err < tolerance || last)
                break;
        }
    }
    GM_HIP(hipMemcpyAsync(scores_out, scores.p, n * 4, hipMemcpyDeviceToHost, st));
    GM_HIP(hipStreamSynchronize(st));
    *iterations_out = iter;
    *error_out = err;
    return GM_OK;
}

GM_API int gm_page_rank(const gm_csr *in_csr, const uint32_t *out_degree, uint64_t max_iterations, double tolerance,
                        float damping_factor, int mode, float *scores_out, uint64_t *iterations_out,
                        double *error_out)
{
    return page_rank_impl(in_csr, out_degree, nullptr, max_iterations, tolerance, damping_factor, mode, scores_out,
                          iterations_out, error_out);
}

GM_API int gm_page_rank_directed(const gm_csr *out_csr, const gm_csr *in_csr, uint64_t max_iterations, double tolerance,
                                 float damping_factor, int mode, float *scores_out, uint64_t *iterations_out,
                                 double *error_out)
{
    GM_CHECK(out_csr, GM_ERR_INVALID, "gm_page_rank_directed: null out-CSR");
    return page_rank_impl(in_csr, nullptr, out_csr, max_iterations, tolerance, damping_factor, mode, scores_out,
                          iterations_out, error_out);
}

namespace gm {
void warm_pagerank() // (common.hpp: the code object of this file, loaded ahead of an algorithm's first call)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&pr_init_kernel)) != hipSuccess)
        (void)hipGetLastError();
}
} // namespace gm
