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
//     flagged node whose bucket (u32)(d/delta) (sssp.rs:192) is <= the current bucket and carries
//     the others over, tracking the minimum far bucket — the reference's min_non_empty_bin
//     (sssp.rs:159-168);
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

__device__ __forceinline__ uint32_t bucket_of(float d, float delta)
{
    const float q = __fdiv_rn(d, delta);
    return q >= 4294967040.0f ? 0xFFFFFFFEu : (uint32_t)q; // saturating, like Rust's `as usize`
}

struct RelaxOut {
    uint32_t again;
    uint32_t far;
};

__device__ __forceinline__ void relax_edge(uint32_t *dist, uint8_t *__restrict__ flag_next, float du, uint32_t t,
                                           float wt, uint32_t cur, float delta, RelaxOut &ro)
{
    const float nd = __fadd_rn(du, wt);
    const uint32_t nb = __float_as_uint(nd);
    if (nb < ld_agent(&dist[t])) { // cheap pre-check, then the real atomic
        const uint32_t old = atomicMin(&dist[t], nb);
        if (nb < old) {
            flag_next[t] = 1;
            const uint32_t b = bucket_of(nd, delta);
            if (b <= cur)
                ro.again = 1;
            else
                ro.far = b < ro.far ? b : ro.far;
        }
    }
}

// One round.  A wavefront takes 256 consecutive nodes at a time: a 4-byte flag load per lane decides
// whether anything in the group is flagged (in the long tail of rounds almost nothing is), then each
// 64-node quarter is relaxed one lane per node.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_round_kernel(
    const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t *dist,
    uint8_t *__restrict__ flag_cur, uint8_t *__restrict__ flag_next, uint32_t n, uint32_t cur, float delta,
    uint32_t *__restrict__ ctrl /* [0] again, [1] min far bucket */)
{
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
                du = __uint_as_float(ld_agent(&dist[u]));
                const uint32_t b = bucket_of(du, delta);
                if (b <= cur) {
                    s = off[u];
                    e = off[u + 1];
                } else { // not yet its turn: carry over
                    flag_next[u] = 1;
                    ro.far = b < ro.far ? b : ro.far;
                }
            }
            // short lists by their own lane, lists longer than 32 edges by the whole wavefront (measured:
            // an edge-balanced expansion with a shuffle search per edge was 25 % slower — the round is bound
            // by the random dist[] accesses, not by the target stream)
            const uint32_t len = e - s;
            if (len <= SSSP_COOP)
                for (uint32_t i = s; i < e; ++i)
                    relax_edge(dist, flag_next, du, tgt[i], w[i], cur, delta, ro);
            uint64_t big = __ballot(len > SSSP_COOP);
            while (big) {
                const int src = __ffsll((unsigned long long)big) - 1;
                big &= big - 1;
                const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
                const float bd = __shfl(du, src, kWave);
                for (uint32_t i = bs + lane; i < be; i += kWave)
                    relax_edge(dist, flag_next, bd, tgt[i], w[i], cur, delta, ro);
            }
        }
    }
    const uint32_t far = wave_min(ro.far);
    const uint64_t any = __ballot(ro.again != 0);
    if (lane == 0) {
        if (any)
            atomicOr(&ctrl[0], 1u);
        if (far != NO_BUCKET)
            atomicMin(&ctrl[1], far);
    }
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
    GM_TRY(ctrl.alloc(16));
    GM_TRY(hctrl.alloc(16));
    hipStream_t st = 0;
    unsigned grid = gm::div_up(n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;

    GM_HIP(hipMemsetAsync(ctrl.p, 0, 16, st));
    if (g->m) {
        unsigned wg = gm::div_up(g->m, 256);
        hipLaunchKernelGGL(sssp_check_weights_kernel, dim3(wg > 8192 ? 8192 : wg), dim3(256), 0, st, g->weights, g->m,
                           ctrl.as<uint32_t>() + 2);
    }
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, dist.as<uint32_t>(), n,
                       (uint32_t)start_node);
    GM_HIP(hipMemsetAsync(flags.p, 0, n_flags * 2, st));
    uint8_t *fcur = flags.as<uint8_t>(), *fnext = flags.as<uint8_t>() + n_flags;
    GM_HIP(hipMemsetAsync(fcur + start_node, 1, 1, st));
    GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 16, hipMemcpyDeviceToHost, st));
    GM_HIP(hipStreamSynchronize(st));
    GM_CHECK(hctrl.as<uint32_t>()[2] == 0, GM_ERR_UNSUPPORTED,
             "gm_sssp_delta_stepping: negative or NaN edge weight (the reference assumes weights >= 0)");

    uint32_t cur = 0;
    const uint32_t reset[2] = {0u, NO_BUCKET};
    uint64_t rounds = 0, bucket_moves = 0;
    for (;;) {
        ++rounds;
        GM_HIP(hipMemcpyAsync(ctrl.p, reset, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sssp_round_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, g->offsets, g->targets, g->weights,
                           dist.as<uint32_t>(), fcur, fnext, n, cur, delta, ctrl.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, 8, hipMemcpyDeviceToHost, st));
        GM_HIP(hipStreamSynchronize(st));
        uint8_t *tmp = fcur;
        fcur = fnext;
        fnext = tmp;
        const uint32_t again = hctrl.as<uint32_t>()[0], far = hctrl.as<uint32_t>()[1];
        if (again)
            continue;
        if (far == NO_BUCKET)
            break;
        cur = far;
        ++bucket_moves;
    }
    if (getenv("GM_SSSP_STATS"))
        fprintf(stderr, "sssp: %llu rounds, %llu bucket advances, last bucket %u\n", (unsigned long long)rounds,
                (unsigned long long)bucket_moves, cur);
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
