// sssp.hip — delta-stepping single-source shortest paths on a device-resident weighted out-CSR.
//
// Replaces crates/algos/src/sssp.rs:38-204.  The reference's result is the least fixed point of
// d[v] = min(d[u] (+) w(u,v)) under f32 round-to-nearest addition, which is monotone — so the
// distances do not depend on the relaxation schedule and this kernel is bit-exact with the
// reference although its buckets are organised differently:
//   * distances are kept as u32 bit patterns (non-negative f32 order == unsigned order) and
//     relaxed with atomicMin — the reference's CAS-min loop (sssp.rs:180-202) in one instruction;
//   * instead of per-thread bins copied into a shared frontier (sssp.rs:85-94) there are two
//     byte-flag arrays (double-buffered per round, so no in-round ordering is needed): a node is
//     flagged when its distance improved since its edges were last relaxed; a round relaxes every
//     flagged node whose distance is <= the current threshold and carries the others over,
//     tracking the minimum pending distance — the role of the reference's min_non_empty_bin
//     (sssp.rs:159-168).  The threshold advances to (minimum pending distance + width) when a round
//     leaves nothing below it; width = delta * GM_SSSP_WIDTH (default 1): `delta` only shapes the
//     schedule, never the result (measured at RMAT scale 24, delta 0.1: widths from delta/16 to pure
//     Bellman-Ford all take 74-98 ms — the small-world graph needs ~7 near-full passes over the edges
//     whatever the order, and finer steps only add rounds);
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
__device__ __forceinline__ void relax_checked(uint32_t *dist, uint8_t *__restrict__ flag_next, uint32_t nb, uint32_t pre,
                                              uint32_t t, uint32_t thr, RelaxOut &ro)
{
    if (nb < pre) {
        const uint32_t old = atomicMin(&dist[t], nb);
        if (nb < old) {
            flag_next[t] = 1;
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
                                            uint8_t *__restrict__ flag_next, float du, uint32_t first, uint32_t end,
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
            relax_checked(dist, flag_next, nb[k], pre[k], t[k], thr, ro);
    }
}

// ctrl words shared by the round and advance kernels
enum : uint32_t { C_AGAIN = 0, C_FAR = 1, C_BAD = 2, C_THR = 3, C_DONE = 4, C_ROUND = 5, C_ADVANCES = 6 };

// One round.  A wavefront takes 256 consecutive nodes at a time: a 4-byte flag load per lane decides
// whether anything in the group is flagged (in the long tail of rounds almost nothing is), then each
// 64-node quarter is relaxed one lane per node.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_round_kernel(
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t *dist,
    uint8_t *__restrict__ flags, size_t n_flags, uint32_t n, uint32_t *ctrl)
{
    if (ld_agent(&ctrl[C_DONE]))
        return; // a round enqueued behind the last one of its batch
    const uint32_t thr = ld_agent(&ctrl[C_THR]);
    const uint32_t parity = ld_agent(&ctrl[C_ROUND]) & 1u;
    uint8_t *__restrict__ flag_cur = flags + (parity ? n_flags : 0);
    uint8_t *__restrict__ flag_next = flags + (parity ? 0 : n_flags);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t ngroups = (n + 255u) >> 8; // the flag arrays are padded to a multiple of 256 bytes
    const uint32_t *flag_words = reinterpret_cast<const uint32_t *>(flag_cur);
    RelaxOut ro{0u, NO_BUCKET};
    for (uint32_t grp = wave; grp < ngroups; grp += nwaves) {
        const uint32_t word = flag_words[grp * 64u + lane];
        if (__ballot(word != 0u) == 0ull)
            continue;
        for (uint32_t quarter = 0; quarter < 4; ++quarter) {
            // lanes whose word covers this quarter: bytes [quarter*64, quarter*64+64) = words quarter*16 .. +16
            const uint32_t u = (grp << 8) + quarter * 64u + lane;
            uint32_t s = 0, e = 0;
            float du = 0.0f;
            if (u < n && flag_cur[u]) {
                flag_cur[u] = 0; // this lane is the only reader/writer of flag_cur[u] in this round
                const uint32_t db = ld_agent(&dist[u]);
                du = __uint_as_float(db);
                if (db <= thr) {
                    s = off[u];
                    e = off[u + 1];
                } else { // not yet its turn: carry over
                    flag_next[u] = 1;
                    ro.far = db < ro.far ? db : ro.far;
                }
            }
            // short lists by their own lane, lists longer than 32 edges by the whole wavefront (measured:
            // an edge-balanced expansion with a shuffle search per edge was 25 % slower — the round is bound
            // by the random dist[] accesses, not by the target stream)
            const uint32_t len = e - s;
            if (len <= SSSP_COOP)
                relax_range(tgt, w, dist, flag_next, du, s, e, 1u, thr, ro);
            uint64_t big = __ballot(len > SSSP_COOP);
            while (big) {
                const int src = __ffsll((unsigned long long)big) - 1;
                big &= big - 1;
                const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
                const float bd = __shfl(du, src, kWave);
                relax_range(tgt, w, dist, flag_next, bd, bs + lane, be, kWave, thr, ro);
            }
        }
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
// distance + width, or finish; then swap the flag buffers (round parity) and clear the round's outputs.
__global__ void sssp_advance_kernel(uint32_t *ctrl, float width)
{
    if (ctrl[C_DONE])
        return;
    if (!ctrl[C_AGAIN]) {
        if (ctrl[C_FAR] == NO_BUCKET) {
            ctrl[C_DONE] = 1u;
        } else {
            const float next = __fadd_rn(__uint_as_float(ctrl[C_FAR]), width);
            const uint32_t nb = __float_as_uint(next);
            ctrl[C_THR] = nb > ctrl[C_FAR] && next < 3.0e38f ? nb : ctrl[C_FAR]; // always covers the pending minimum
            ctrl[C_ADVANCES] += 1u;
        }
    }
    ctrl[C_AGAIN] = 0u;
    ctrl[C_FAR] = NO_BUCKET;
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
    const size_t n_flags = ((size_t)n + 255) & ~(size_t)255; // the round kernel reads flags 4 bytes per lane
    GM_TRY(flags.alloc(n_flags * 2));
    GM_TRY(ctrl.alloc(32));
    GM_TRY(hctrl.alloc(32));
    hipStream_t st = 0;
    unsigned grid = gm::div_up(n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    float frac = 1.0f;
    if (const char *v = getenv("GM_SSSP_WIDTH"))
        frac = (float)atof(v);
    if (!(frac > 0.0f))
        frac = 1.0f;
    const float width = delta * frac;

    // ctrl: again 0, far NONE, bad 0, threshold 0.0 (only the start node qualifies), done 0, round 0, advances 0
    const uint32_t init_ctrl[8] = {0u, NO_BUCKET, 0u, 0u, 0u, 0u, 0u, 0u};
    GM_HIP(hipMemcpyAsync(ctrl.p, init_ctrl, 32, hipMemcpyHostToDevice, st));
    if (g->m) {
        unsigned wg = gm::div_up(g->m, 256);
        hipLaunchKernelGGL(sssp_check_weights_kernel, dim3(wg > 8192 ? 8192 : wg), dim3(256), 0, st, g->weights, g->m,
                           ctrl.as<uint32_t>() + C_BAD);
    }
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, dist.as<uint32_t>(), n,
                       (uint32_t)start_node);
    GM_HIP(hipMemsetAsync(flags.p, 0, n_flags * 2, st));
    GM_HIP(hipMemsetAsync(flags.as<uint8_t>() + start_node, 1, 1, st));
    GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 32, hipMemcpyDeviceToHost, st));
    GM_HIP(hipStreamSynchronize(st));
    GM_CHECK(hctrl.as<uint32_t>()[C_BAD] == 0, GM_ERR_UNSUPPORTED,
             "gm_sssp_delta_stepping: negative or NaN edge weight (the reference assumes weights >= 0)");

    const bool stats = getenv("GM_SSSP_STATS") != nullptr;
    const int batch = stats ? 1 : 8; // rounds enqueued per host synchronisation
    auto t_prev = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(sssp_round_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, g->offsets, g->targets, g->weights,
                               dist.as<uint32_t>(), flags.as<uint8_t>(), n_flags, n, ctrl.as<uint32_t>());
            hipLaunchKernelGGL(sssp_advance_kernel, dim3(1), dim3(1), 0, st, ctrl.as<uint32_t>(), width);
        }
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 32, hipMemcpyDeviceToHost, st));
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
        fprintf(stderr, "sssp: %u rounds, %u threshold advances, width %.6f\n", hctrl.as<uint32_t>()[C_ROUND],
                hctrl.as<uint32_t>()[C_ADVANCES], width);
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
