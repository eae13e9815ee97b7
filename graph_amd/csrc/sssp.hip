// sssp.hip — delta-stepping single-source shortest paths on a device-resident weighted out-CSR.
//
// Replaces crates/algos/src/sssp.rs:38-204.  The reference's intended result is the least fixed point of
// d[v] = min(d[u] (+) w(u,v)) under f32 round-to-nearest addition, which is monotone — so the
// distances do not depend on the relaxation schedule and this kernel returns exactly that fixed point,
// bit for bit, although its buckets are organised differently.  (The reference itself misses the fixed
// point on inputs where its stale-entry test `d >= delta * bin` (:126) disagrees in f32 with the bin it
// chose as (d/delta) as usize (:192), e.g. d = 13.5, delta = 0.3: the node is skipped and its edges are
// never relaxed.  There is no such test here)
//   * distances are kept as u32 bit patterns (non-negative f32 order == unsigned order) and
//     relaxed with atomicMin — the reference's CAS-min loop (sssp.rs:180-202) in one instruction;
//   * instead of per-thread bins copied into a shared frontier (sssp.rs:85-94) there is one bit per
//     node: set when the node's distance improved since its edges were last relaxed.  A round clears
//     and relaxes every flagged node whose distance is <= the current threshold; the others simply
//     stay flagged (nothing is rewritten for them), and the minimum pending distance is tracked — the
//     role of the reference's min_non_empty_bin (sssp.rs:159-168).  No fences: "distance improved,
//     then flag set" and "flag cleared, then distance read" are each ordered by the returned value of
//     the first atomic, so an improvement is never lost;  The threshold advances to (minimum pending distance + width) when a round
//     leaves nothing below it; `delta` only seeds the schedule (first step delta/32, then adapted to
//     the work of each phase, see gm_sssp_delta_stepping) and never changes the result;
//   * the bucket bookkeeping lives on the device (sssp_advance_kernel): the host enqueues rounds in
//     batches and reads one flag per batch instead of synchronising after every round;
//   * INF = f32::MAX (sssp.rs:12), never +inf.
// One lane per node; adjacency lists longer than 32 edges are relaxed by the whole wavefront.
#include "common.hpp"
#include "device_utils.hpp"

#include <cstdlib>

namespace {

using namespace gm;

constexpr int SSSP_BLOCK = 256;
constexpr uint32_t SSSP_COOP = 32; // lists longer than this are relaxed by the whole wavefront
constexpr uint32_t SSSP_INF_BITS = 0x7F7FFFFFu; // f32::MAX
constexpr uint32_t NO_BUCKET = 0xFFFFFFFFu;

struct RelaxOut {
    uint32_t again;
    uint32_t far;
};

// thr: bit pattern of the current distance threshold (non-negative f32 order == unsigned order).
// `pre` is dist[t] as read by the caller's batched pre-check (several independent random reads in
// flight per lane); the atomic only runs when the candidate still looks like an improvement.
__device__ __forceinline__ void relax_checked(uint32_t *dist, uint32_t *flags, uint32_t *wmin, uint32_t nb, uint32_t pre,
                                              uint32_t t, uint32_t thr, RelaxOut &ro)
{
    if (nb < pre) {
        const uint32_t old = atomicMin(&dist[t], nb);
        if (nb < old) { // issued only once the atomicMin has returned: distance, then flag, then word summary
            uint32_t zero = atomicOr(&flags[t >> 5], 1u << (t & 31u));
            asm volatile("v_and_b32 %0, 0, %0" : "+v"(zero)); // 0, but only known once the atomicOr has returned
            atomicMin(&wmin[(t >> 5) + zero], nb);
            if (nb <= thr)
                ro.again = 1;
            else
                ro.far = nb < ro.far ? nb : ro.far;
        }
    }
}

// relaxes edges i = first, first + step, ... < end of one source at distance du, SSSP_MLP at a time
constexpr int SSSP_MLP = 4;
__device__ __forceinline__ void relax_range(const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t *dist,
                                            uint32_t *flags, uint32_t *wmin, float du, uint32_t first, uint32_t end,
                                            uint32_t step, uint32_t thr, RelaxOut &ro)
{
    for (uint32_t i = first; i < end; i += step * SSSP_MLP) {
        uint32_t t[SSSP_MLP], nb[SSSP_MLP], pre[SSSP_MLP];
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k) {
            const uint32_t j = i + (uint32_t)k * step;
            const bool in = j < end;
            t[k] = in ? tgt[j] : 0u;
            nb[k] = in ? __float_as_uint(__fadd_rn(du, w[j])) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            pre[k] = nb[k] != 0xFFFFFFFFu ? ld_agent(&dist[t[k]]) : 0u;
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            relax_checked(dist, flags, wmin, nb[k], pre[k], t[k], thr, ro);
    }
}

// ctrl words shared by the round and advance kernels
enum : uint32_t { C_AGAIN = 0, C_FAR = 1, C_BAD = 2, C_THR = 3, C_DONE = 4, C_ROUND = 5, C_ADVANCES = 6,
                  C_WORK = 7 /* relaxed edges / 64, statistics */, C_CHUNKS = 8 /* deferred edge chunks of this round */,
                  C_WIDTH = 9 /* f32 bits: current threshold step */, C_MARK = 10 /* C_WORK at the last advance */ };

// One round.  wmin[i] is a lower bound of the distances of the flagged nodes of flag word i (32 nodes):
// whoever flags a node lowers it, the scanner resets it and puts back what it leaves flagged.  A
// wavefront takes 64 words (2048 consecutive nodes) at a time with one coalesced wmin load; only words
// that can hold a node at or below the threshold are opened (two at a time, one per half wavefront),
// their near nodes cleared and collected in an LDS list, then relaxed one lane per node (lists longer
// than 32 edges by the whole wavefront).  Nodes above the threshold cost nothing per round.
constexpr uint32_t SSSP_GROUP = 64u * 32u; // nodes per wavefront step
constexpr uint32_t SSSP_BIG = 2048;        // adjacency lists longer than this are cut into chunks for the whole grid
constexpr uint32_t SSSP_CHUNK = 1024;      // edges per deferred chunk

__device__ __forceinline__ void sssp_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(SSSP_BLOCK) void sssp_round_kernel(const uint32_t *__restrict__ off,
                                                                const uint32_t *__restrict__ tgt,
                                                                const float *__restrict__ w, uint32_t *dist,
                                                                uint32_t *flags, uint32_t *wmin, uint32_t nwords,
                                                                uint2 *__restrict__ chunks, uint32_t *ctrl)
{
    __shared__ uint32_t list[SSSP_BLOCK / kWave][SSSP_GROUP];
    if (ld_agent(&ctrl[C_DONE]))
        return; // a round enqueued behind the last one of its batch
    const uint32_t thr = ld_agent(&ctrl[C_THR]);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t half = lane & 32u, sub = lane & 31u;
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t ngroups = (nwords + kWave - 1) / kWave;
    const uint64_t lt_mask = lane ? (~0ull >> (64u - lane)) : 0ull;
    RelaxOut ro{0u, NO_BUCKET};
    uint32_t work = 0; // out-edges of the nodes this lane relaxed (statistics)
    for (uint32_t grp = wave; grp < ngroups; grp += nwaves) {
        const uint32_t my_word = grp * kWave + lane;
        const uint32_t lo = my_word < nwords ? ld_agent(&wmin[my_word]) : NO_BUCKET;
        if (lo > thr && lo != NO_BUCKET)
            ro.far = lo < ro.far ? lo : ro.far;
        uint64_t cand = __ballot(lo <= thr);
        uint32_t total = 0;
        while (cand) {
            // two words per step: the lower half of the wavefront opens the first, the upper half the second
            const uint32_t i0 = (uint32_t)__ffsll((unsigned long long)cand) - 1u;
            cand &= cand - 1;
            uint32_t i1 = 64u;
            if (cand) {
                i1 = (uint32_t)__ffsll((unsigned long long)cand) - 1u;
                cand &= cand - 1;
            }
            const uint32_t mine = half ? i1 : i0;
            const bool open = mine < 64u;
            const uint32_t widx = grp * kWave + (open ? mine : 0u);
            if (open && sub == 0)
                atomicExch(&wmin[widx], NO_BUCKET); // reset first, then look: a concurrent flagger re-arms the word
            sssp_drain();
            const uint32_t fw = open ? ld_agent(&flags[widx]) : 0u;
            const bool bit = (fw >> sub) & 1u;
            const uint32_t u = widx * 32u + sub;
            const uint32_t db = bit ? ld_agent(&dist[u]) : NO_BUCKET;
            const bool is_near = bit && db <= thr;
            uint32_t keep_far = bit && !is_near ? db : NO_BUCKET; // what stays flagged goes back into the summary
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const uint32_t other = __shfl_xor(keep_far, o, kWave);
                keep_far = other < keep_far ? other : keep_far;
            }
            if (keep_far != NO_BUCKET) {
                if (sub == 0)
                    atomicMin(&wmin[widx], keep_far);
                ro.far = keep_far < ro.far ? keep_far : ro.far;
            }
            const uint64_t near_b = __ballot(is_near);
            const uint32_t clear = half ? (uint32_t)(near_b >> 32) : (uint32_t)near_b;
            if (clear && sub == 0)
                atomicAnd(&flags[widx], ~clear);
            sssp_drain(); // cleared before anyone reads the distances the relaxation uses
            if (is_near)
                list[wv][total + (uint32_t)__popcll(near_b & lt_mask)] = u;
            total += (uint32_t)__popcll(near_b);
        }
        if (total == 0)
            continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t base = 0; base < total; base += kWave) {
            uint32_t s = 0, e = 0;
            float du = 0.0f;
            if (base + lane < total) {
                const uint32_t u = list[wv][base + lane];
                du = __uint_as_float(ld_agent(&dist[u])); // <= thr: distances only decrease
                s = off[u];
                e = off[u + 1];
            }
            // short lists by their own lane, lists longer than 32 edges by the whole wavefront (measured:
            // an edge-balanced expansion with a shuffle search per edge was 25 % slower — the round is bound
            // by the random dist[] accesses, not by the target stream)
            const uint32_t len = e - s;
            work += len;
            if (len <= SSSP_COOP)
                relax_range(tgt, w, dist, flags, wmin, du, s, e, 1u, thr, ro);
            uint64_t big = __ballot(len > SSSP_COOP && len <= SSSP_BIG);
            while (big) {
                const int src = __ffsll((unsigned long long)big) - 1;
                big &= big - 1;
                const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
                const float bd = __shfl(du, src, kWave);
                relax_range(tgt, w, dist, flags, wmin, bd, bs + lane, be, kWave, thr, ro);
            }
            // hubs: one wavefront would be the round's critical path (a 300k-edge list = milliseconds);
            // their edge ranges go to a chunk list that sssp_chunk_kernel spreads over the whole grid
            uint64_t huge = __ballot(len > SSSP_BIG);
            while (huge) {
                const int src = __ffsll((unsigned long long)huge) - 1;
                huge &= huge - 1;
                const uint32_t hu = __shfl(base + lane < total ? list[wv][base + lane] : 0u, src, kWave);
                const uint32_t hs = __shfl(s, src, kWave), he = __shfl(e, src, kWave);
                const uint32_t nch = (he - hs + SSSP_CHUNK - 1u) / SSSP_CHUNK;
                uint32_t first = 0;
                if (lane == 0)
                    first = atomicAdd(&ctrl[C_CHUNKS], nch);
                first = __shfl(first, 0, kWave);
                for (uint32_t c = lane; c < nch; c += kWave)
                    chunks[first + c] = make_uint2(hu, hs + c * SSSP_CHUNK);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the list is rewritten by the next step
    }
    const uint32_t far = wave_min(ro.far);
    const uint64_t any = __ballot(ro.again != 0);
    const uint32_t wave_work = (uint32_t)wave_sum((uint64_t)work);
    if (lane == 0) {
        if (wave_work)
            atomicAdd(&ctrl[C_WORK], (wave_work + 63u) >> 6);
        if (any)
            atomicOr(&ctrl[C_AGAIN], 1u);
        if (far != NO_BUCKET)
            atomicMin(&ctrl[C_FAR], far);
    }
}

// The deferred hub edges of the round that just ran: chunk (u, first edge) -> up to SSSP_CHUNK edges of u,
// one wavefront per chunk, any wavefront of the grid.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_chunk_kernel(const uint32_t *__restrict__ off,
                                                                const uint32_t *__restrict__ tgt,
                                                                const float *__restrict__ w, uint32_t *dist,
                                                                uint32_t *flags, uint32_t *wmin,
                                                                const uint2 *__restrict__ chunks, uint32_t *ctrl)
{
    if (ld_agent(&ctrl[C_DONE]))
        return;
    const uint32_t nchunks = ld_agent(&ctrl[C_CHUNKS]);
    if (nchunks == 0)
        return;
    const uint32_t thr = ld_agent(&ctrl[C_THR]);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    RelaxOut ro{0u, NO_BUCKET};
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint2 ch = chunks[c];
        const float du = __uint_as_float(ld_agent(&dist[ch.x]));
        const uint32_t end_u = off[ch.x + 1];
        const uint32_t end = ch.y + SSSP_CHUNK < end_u ? ch.y + SSSP_CHUNK : end_u;
        relax_range(tgt, w, dist, flags, wmin, du, ch.y + lane, end, kWave, thr, ro);
    }
    const uint32_t far = wave_min(ro.far);
    const uint64_t any = __ballot(ro.again != 0);
    if (lane == 0) {
        if (any)
            atomicOr(&ctrl[C_AGAIN], 1u);
        if (far != NO_BUCKET)
            atomicMin(&ctrl[C_FAR], far);
    }
}

// Between two rounds (one thread): nothing left below the threshold -> move it to the minimum pending
// distance + width, or finish; then clear the round's outputs.
// adapt_lo / adapt_hi (units of 64 relaxed edges; 0 = fixed width): the step halves when the phase that
// just ended relaxed more than adapt_hi and doubles when it relaxed less than adapt_lo, within
// [width_min, width_max] — coarse steps re-relax every edge ~6 times, fine steps leave the chip idle.
__global__ void sssp_advance_kernel(uint32_t *ctrl, uint32_t adapt_lo, uint32_t adapt_hi, float width_min, float width_max)
{
    if (ctrl[C_DONE])
        return;
    if (!ctrl[C_AGAIN]) {
        if (ctrl[C_FAR] == NO_BUCKET) {
            ctrl[C_DONE] = 1u;
        } else {
            float width = __uint_as_float(ctrl[C_WIDTH]);
            if (adapt_hi) {
                const uint32_t phase = ctrl[C_WORK] - ctrl[C_MARK];
                if (phase > adapt_hi)
                    width = fmaxf(width * 0.5f, width_min);
                else if (phase < adapt_lo)
                    width = fminf(width * 2.0f, width_max);
                ctrl[C_WIDTH] = __float_as_uint(width);
                ctrl[C_MARK] = ctrl[C_WORK];
            }
            const float next = __fadd_rn(__uint_as_float(ctrl[C_FAR]), width);
            const uint32_t nb = __float_as_uint(next);
            ctrl[C_THR] = nb > ctrl[C_FAR] && next < 3.0e38f ? nb : ctrl[C_FAR]; // always covers the pending minimum
            ctrl[C_ADVANCES] += 1u;
        }
    }
    ctrl[C_AGAIN] = 0u;
    ctrl[C_FAR] = NO_BUCKET;
    ctrl[C_CHUNKS] = 0u;
    ctrl[C_ROUND] += 1u;
}

// Partitioned building block: relax every out-edge of the slice's rows whose distance is finite
// (row r of the slice is node row_base + r).  changed[0] is set when any distance improved.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_relax_rows_kernel(const uint32_t *__restrict__ off,
                                                                     const uint32_t *__restrict__ tgt,
                                                                     const float *__restrict__ w, uint32_t *dist,
                                                                     uint32_t rows, uint32_t row_base,
                                                                     uint32_t *__restrict__ changed)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rows_pad = (rows + kWave - 1) / kWave * kWave;
    bool improved = false;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows_pad; r += stride) {
        uint32_t s = 0, e = 0;
        float du = 0.0f;
        if (r < rows) {
            const uint32_t bits = ld_agent(&dist[row_base + r]);
            if (bits != SSSP_INF_BITS) {
                du = __uint_as_float(bits);
                s = off[r];
                e = off[r + 1];
            }
        }
        const uint32_t len = e - s;
        if (len <= SSSP_COOP)
            for (uint32_t i = s; i < e; ++i) {
                const uint32_t nb = __float_as_uint(__fadd_rn(du, w[i]));
                if (nb < ld_agent(&dist[tgt[i]]) && nb < atomicMin(&dist[tgt[i]], nb))
                    improved = true;
            }
        uint64_t big = __ballot(len > SSSP_COOP);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
            const float bd = __shfl(du, src, kWave);
            for (uint32_t i = bs + lane; i < be; i += kWave) {
                const uint32_t nb = __float_as_uint(__fadd_rn(bd, w[i]));
                if (nb < ld_agent(&dist[tgt[i]]) && nb < atomicMin(&dist[tgt[i]], nb))
                    improved = true;
            }
        }
    }
    if (__ballot(improved) && lane == 0)
        atomicOr(changed, 1u);
}

__global__ void sssp_init_kernel(uint32_t *__restrict__ dist, uint32_t n, uint32_t start)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride)
        dist[u] = u == start ? 0u : SSSP_INF_BITS;
}

__global__ void sssp_check_weights_kernel(const float *__restrict__ w, uint64_t m, uint32_t *__restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        if (!(w[i] >= 0.0f))
            *bad = 1;
}

} // namespace

GM_API int gm_sssp_delta_stepping(const gm_csr *g, uint64_t start_node, float delta, float *distances_out)
{
    GM_CHECK(g, GM_ERR_INVALID, "gm_sssp_delta_stepping: null CSR");
    GM_CHECK(g->weights || g->m == 0, GM_ERR_INVALID, "gm_sssp_delta_stepping: the CSR carries no weights");
    // sssp.rs:52: distance[start_node] panics when out of bounds
    GM_CHECK(start_node < g->n, GM_ERR_RANGE, "gm_sssp_delta_stepping: start_node %llu >= node_count %llu",
             (unsigned long long)start_node, (unsigned long long)g->n);
    GM_CHECK(delta > 0.0f && delta < 3.0e38f, GM_ERR_INVALID,
             "gm_sssp_delta_stepping: delta must be a positive finite f32 (reference: bin index overflow panic)");
    GM_CHECK(distances_out, GM_ERR_INVALID, "gm_sssp_delta_stepping: distances_out is null");
    gm::DeviceGuard guard(g->device);
    const uint32_t n = (uint32_t)g->n;
    gm::DevBuf dist, flags, ctrl;
    gm::PinnedBuf hctrl;
    GM_TRY(dist.alloc((size_t)n * 4));
    const uint32_t nwords = (n + 31u) / 32u; // one flag bit per node, one distance summary per flag word
    GM_TRY(flags.alloc(((size_t)nwords + kWave) * 4));
    gm::DevBuf wmin;
    GM_TRY(wmin.alloc(((size_t)nwords + kWave) * 4));
    GM_TRY(ctrl.alloc(64));
    GM_TRY(hctrl.alloc(64));
    gm::DevBuf chunks; // (node, first edge) of deferred hub chunks: at most one per SSSP_CHUNK edges plus one per node
    GM_TRY(chunks.alloc(((size_t)g->m / SSSP_CHUNK + (size_t)g->m / SSSP_BIG + 64) * sizeof(uint2)));
    hipStream_t st = 0;
    unsigned grid = gm::div_up(n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    // Threshold step: starts at delta/32 and adapts to the work of each phase (sssp_advance_kernel): it doubles
    // while a phase relaxes fewer than m/5 edges and halves beyond 3m/4.  Measured at RMAT scale 24, delta 0.1:
    // 2.0 x m relaxations in ~60 rounds, 32 ms; a fixed step of delta: 6.4 x m, 53 ms; fixed delta/16: 2.2 x m but
    // 500 rounds, 101 ms.  GM_SSSP_WIDTH=<fraction of delta> sets the first step, GM_SSSP_ADAPT="lo,hi" the band
    // in millions of edges ("0,0": fixed step).
    float frac = 1.0f / 32.0f;
    if (const char *v = getenv("GM_SSSP_WIDTH"))
        frac = (float)atof(v);
    if (!(frac > 0.0f))
        frac = 1.0f / 32.0f;
    const float width = delta * frac;
    uint32_t adapt_lo = (uint32_t)(g->m / 5 / 64), adapt_hi = (uint32_t)(g->m / 4 * 3 / 64) + 1u;
    if (const char *v = getenv("GM_SSSP_ADAPT")) {
        double lo = 0, hi = 0;
        if (sscanf(v, "%lf,%lf", &lo, &hi) == 2 && lo >= 0) {
            adapt_lo = hi > lo ? (uint32_t)(lo * 1e6 / 64.0) : 0u;
            adapt_hi = hi > lo ? (uint32_t)(hi * 1e6 / 64.0) : 0u;
        }
    }

    // ctrl: again 0, far NONE, bad 0, threshold 0.0 (only the start node qualifies), done 0, round 0, advances 0
    uint32_t init_ctrl[16] = {0u, NO_BUCKET, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    memcpy(&init_ctrl[C_WIDTH], &width, 4);
    GM_HIP(hipMemcpyAsync(ctrl.p, init_ctrl, 64, hipMemcpyHostToDevice, st));
    if (g->m) {
        unsigned wg = gm::div_up(g->m, 256);
        hipLaunchKernelGGL(sssp_check_weights_kernel, dim3(wg > 8192 ? 8192 : wg), dim3(256), 0, st, g->weights, g->m,
                           ctrl.as<uint32_t>() + C_BAD);
    }
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, dist.as<uint32_t>(), n,
                       (uint32_t)start_node);
    GM_HIP(hipMemsetAsync(flags.p, 0, flags.bytes, st));
    const uint32_t start_bit = 1u << (start_node & 31u);
    GM_HIP(hipMemcpyAsync(flags.as<uint32_t>() + (start_node >> 5), &start_bit, 4, hipMemcpyHostToDevice, st));
    GM_HIP(hipMemsetAsync(wmin.p, 0xFF, wmin.bytes, st));
    GM_HIP(hipMemsetAsync(wmin.as<uint32_t>() + (start_node >> 5), 0, 4, st)); // the start node's distance: 0.0
    GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 64, hipMemcpyDeviceToHost, st));
    GM_HIP(hipStreamSynchronize(st));
    GM_CHECK(hctrl.as<uint32_t>()[C_BAD] == 0, GM_ERR_UNSUPPORTED,
             "gm_sssp_delta_stepping: negative or NaN edge weight (the reference assumes weights >= 0)");

    const bool stats = getenv("GM_SSSP_STATS") != nullptr;
    const int batch = stats ? 1 : 8; // rounds enqueued per host synchronisation
    auto t_prev = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(sssp_round_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, g->offsets, g->targets, g->weights,
                               dist.as<uint32_t>(), flags.as<uint32_t>(), wmin.as<uint32_t>(), nwords, chunks.as<uint2>(),
                               ctrl.as<uint32_t>());
            hipLaunchKernelGGL(sssp_chunk_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, g->offsets, g->targets, g->weights,
                               dist.as<uint32_t>(), flags.as<uint32_t>(), wmin.as<uint32_t>(), chunks.as<uint2>(),
                               ctrl.as<uint32_t>());
            hipLaunchKernelGGL(sssp_advance_kernel, dim3(1), dim3(1), 0, st, ctrl.as<uint32_t>(), adapt_lo, adapt_hi,
                               delta / 1024.0f, delta * 16.0f);
        }
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 64, hipMemcpyDeviceToHost, st));
        GM_HIP(hipStreamSynchronize(st));
        const uint32_t *hc = hctrl.as<uint32_t>();
        if (stats) { // batch = 1: wall clock between synchronisations is the round time
            const auto t_now = std::chrono::steady_clock::now();
            fprintf(stderr, "sssp round %u threshold %.6f: %.3f ms\n", hc[C_ROUND], __builtin_bit_cast(float, hc[C_THR]),
                    std::chrono::duration<double, std::milli>(t_now - t_prev).count());
            t_prev = t_now;
        }
        if (hc[C_DONE])
            break;
    }
    if (stats)
        fprintf(stderr, "sssp: %u rounds, %u threshold advances, width %.6f (last %.6f), ~%.1f M edge relaxations (%.2f x m)\n",
                hctrl.as<uint32_t>()[C_ROUND], hctrl.as<uint32_t>()[C_ADVANCES], width,
                __builtin_bit_cast(float, hctrl.as<uint32_t>()[C_WIDTH]),
                hctrl.as<uint32_t>()[C_WORK] * 64.0 / 1e6, g->m ? hctrl.as<uint32_t>()[C_WORK] * 64.0 / (double)g->m : 0.0);
    GM_HIP(hipMemcpy(distances_out, dist.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GM_OK;
}


// ------------------------------------------------------------------------------------------------
// Building blocks of the partitioned run (distances replicated as u32 bit patterns, every
// rank relaxes the out-edges of its own rows, min-all-reduce between rounds until nothing changes).
// The result is the same least fixed point as gm_sssp_delta_stepping — schedule-free.
// ------------------------------------------------------------------------------------------------
GM_API int gm_sssp_init_distances(uint64_t n, uint64_t start_node, uint64_t d_dist_bits, int device, void *stream)
{
    GM_CHECK(d_dist_bits, GM_ERR_INVALID, "gm_sssp_init_distances: null distances");
    GM_CHECK(start_node < n && n < (1ull << 32), GM_ERR_RANGE, "gm_sssp_init_distances: start_node %llu >= node_count %llu",
             (unsigned long long)start_node, (unsigned long long)n);
    gm::DeviceGuard guard(device);
    unsigned grid = gm::div_up(n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<uint32_t *>(d_dist_bits), (uint32_t)n, (uint32_t)start_node);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_sssp_relax_rows(const gm_csr *out_rows, uint64_t row_begin, uint64_t n_global, uint64_t d_dist_bits,
                              uint64_t d_changed, void *stream)
{
    GM_CHECK(out_rows && d_dist_bits && d_changed, GM_ERR_INVALID, "gm_sssp_relax_rows: null argument");
    GM_CHECK(out_rows->weights || out_rows->m == 0, GM_ERR_INVALID, "gm_sssp_relax_rows: the CSR carries no weights");
    GM_CHECK(row_begin + out_rows->n <= n_global && n_global < (1ull << 32), GM_ERR_RANGE,
             "gm_sssp_relax_rows: rows outside the graph");
    if (out_rows->n == 0)
        return GM_OK;
    gm::DeviceGuard guard(out_rows->device);
    unsigned grid = gm::div_up(out_rows->n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    hipLaunchKernelGGL(sssp_relax_rows_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, (hipStream_t)stream, out_rows->offsets,
                       out_rows->targets, out_rows->weights, reinterpret_cast<uint32_t *>(d_dist_bits),
                       (uint32_t)out_rows->n, (uint32_t)row_begin, reinterpret_cast<uint32_t *>(d_changed));
    GM_HIP(hipGetLastError());
    return GM_OK;
}
