// tc.hip — global triangle count on a device-resident undirected CSR with sorted lists.
//
// Replaces crates/algos/src/triangle_count.rs:22-86 (put-back iterator: crates/algos/src/utils.rs:8-101).
// Restating the reference loop (:47-70): for every entry v of N(u) with v <= u, and every entry w
// of N(v) with w <= v, count 1 iff w occurs in N(u) (the put-back cursor makes the test a pure
// membership test; a duplicate w in N(v) counts again, a duplicate in N(u) does not).  Because
// w <= v <= u, only the lower prefixes L(x) = {y in N(x) : y <= x} matter:
//     triangles = sum over entries v of L(u)  of  #{entries w of L(v) : w in set(L(u))}
// — the "forward" algorithm on the DAG of lower prefixes, exact for Sorted (duplicates and
// self-loops kept: 227874 on the reference's scale_8 fixture after relabelling) and
// Deduplicated layouts alike.
//
// Kernels: tc_low_len (per node: |L(u)| by binary search), tc_order (per entry: sortedness / strictness check),
// tc_dag_src (row id of every DAG entry, filled per node with wavefront help for long lists),
// tc_count (one lane per DAG entry: walk one prefix, binary-search the other; on strictly
// increasing lists the shorter prefix is walked).  Integer work, HBM/latency bound: no MFMA.
#include "common.hpp"
#include "device_utils.hpp"

#include <rocprim/rocprim.hpp>

#include <cstdlib>

namespace {

using namespace gm;

constexpr int TC_BLOCK = 256;
constexpr int TC_WAVES = TC_BLOCK / kWave;

// Per node: |L(u)| by binary search, and a mark on the first entry of its list (row_start bitmap) so
// that the order check below can run one lane per ENTRY: checking a 600k-entry hub list with the one
// wavefront that owns the node was 24 ms of critical path at scale 24.
__global__ __launch_bounds__(TC_BLOCK) void tc_low_len_kernel(const uint32_t *__restrict__ off,
                                                              const uint32_t *__restrict__ tgt, uint32_t n,
                                                              uint32_t *__restrict__ low_len /* n+1 */,
                                                              uint32_t *__restrict__ row_start /* bit per entry */,
                                                              uint32_t *__restrict__ flags)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u <= n; u += stride) {
        if (u == n) {
            low_len[u] = 0;
            continue;
        }
        const uint32_t s = off[u], e = off[u + 1];
        uint32_t lo = s, hi = e; // first position with tgt > u
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (tgt[mid] <= u)
                lo = mid + 1;
            else
                hi = mid;
        }
        low_len[u] = lo - s;
        // a self-loop entry (u in N(u)) takes part in the put-back walk as v == u and as w == v; the bitmap
        // path counts only w < v < u, so lists with a self-loop take the general path (flag 2), like lists
        // with duplicates (an undirected build doubles self-loops; an uploaded CSR may hold a single one)
        if (lo > s && tgt[lo - 1] == u)
            atomicOr(flags, 2u);
        if (e > s)
            atomicOr(&row_start[s >> 5], 1u << (s & 31u));
    }
}

// flags[0] |= 1 if some list is not sorted ascending, |= 2 if some list has equal neighbours.
__global__ __launch_bounds__(TC_BLOCK) void tc_order_kernel(const uint32_t *__restrict__ tgt, uint64_t m,
                                                            const uint32_t *__restrict__ row_start,
                                                            uint32_t *__restrict__ flags)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t f = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < m; i += stride) {
        if ((row_start[i >> 5] >> (i & 31u)) & 1u)
            continue; // tgt[i - 1] belongs to the previous list
        const uint32_t a = tgt[i - 1], b = tgt[i];
        f |= (a > b) ? 1u : 0u;
        f |= (a == b) ? 2u : 0u;
    }
    if (f)
        atomicOr(flags, f);
}

// dag_src[k] = u for k in [loff[u], loff[u+1]); dag_tgt[k] = the k-th lower-prefix entry
__global__ __launch_bounds__(TC_BLOCK) void tc_dag_kernel(const uint32_t *__restrict__ off,
                                                          const uint32_t *__restrict__ tgt,
                                                          const uint32_t *__restrict__ loff, uint32_t n,
                                                          uint32_t *__restrict__ dag_src, uint32_t *__restrict__ dag_tgt)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + kWave - 1) / kWave * kWave;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_pad; u += stride) {
        uint32_t ls = 0, le = 0, s = 0;
        if (u < n) {
            ls = loff[u];
            le = loff[u + 1];
            s = off[u];
        }
        const uint32_t len = le - ls;
        if (len <= 32)
            for (uint32_t i = 0; i < len; ++i) {
                dag_src[ls + i] = u;
                dag_tgt[ls + i] = tgt[s + i];
            }
        uint64_t big = __ballot(len > 32);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bu = __shfl(u, src, kWave), bls = __shfl(ls, src, kWave), ble = __shfl(le, src, kWave),
                           bs = __shfl(s, src, kWave);
            for (uint32_t i = lane; i < ble - bls; i += kWave) {
                dag_src[bls + i] = bu;
                dag_tgt[bls + i] = tgt[bs + i];
            }
        }
    }
}

__device__ __forceinline__ bool tc_contains(const uint32_t *__restrict__ list, uint32_t len, uint32_t x)
{
    uint32_t lo = 0, hi = len;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t y = list[mid];
        if (y < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < len && list[lo] == x;
}

// Triangular adjacency bitmap of the K smallest ids (after make_degree_ordered: the K highest
// degrees): bit (x, y), y < x < K, is set iff y is in L(x).  Row x starts at bit x(x-1)/2.
__device__ __forceinline__ uint64_t tc_bit_index(uint32_t x, uint32_t y) { return (uint64_t)x * (x - 1) / 2 + y; }

__global__ void tc_bitmap_fill_kernel(const uint32_t *__restrict__ dag_src, const uint32_t *__restrict__ dag_tgt,
                                      uint64_t dag_m, uint32_t K, uint32_t *__restrict__ bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < dag_m; k += stride) {
        const uint32_t x = dag_src[k], y = dag_tgt[k];
        if (x < K && y < x) {
            const uint64_t b = tc_bit_index(x, y);
            atomicOr(&bits[b >> 5], 1u << (b & 31));
        }
    }
}

// one lane per DAG entry (u, v): count entries w of L(v) that occur in L(u)
template <bool STRICT>
__global__ __launch_bounds__(TC_BLOCK) void tc_count_kernel(const uint32_t *__restrict__ loff,
                                                            const uint32_t *__restrict__ dag_src,
                                                            const uint32_t *__restrict__ dag_tgt, uint64_t dag_m,
                                                            const uint32_t *__restrict__ bits, uint32_t K,
                                                            unsigned long long *__restrict__ total)
{
    __shared__ uint64_t red[TC_WAVES];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint64_t dag_pad = (dag_m + kWave - 1) / kWave * kWave; // whole wavefronts iterate together
    uint64_t count = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < dag_pad; k += stride) {
        const bool live = k < dag_m;
        const uint32_t u = live ? dag_src[k] : 0u, v = live ? dag_tgt[k] : 0u;
        bool handled = !live;
        if (STRICT) {
            // lists are sets: |L(u) ∩ L(v)| = number of w in L(u), w < v, with bit (v, w) set; the w < v are
            // exactly the entries of L(u) in front of this one — one load per candidate, no search.
            // Short prefixes by the owning lane, long ones spread over the wavefront.
            const bool bitmap = live && v < K && v < u;
            const uint32_t lo_u = bitmap ? loff[u] : 0u;
            const uint32_t rank = bitmap ? (uint32_t)(k - lo_u) : 0u;
            const uint64_t row = (uint64_t)v * (v ? v - 1 : 0) / 2;
            if (bitmap && rank <= 64) {
                const uint32_t *lu = dag_tgt + lo_u;
                uint32_t c = 0;
                for (uint32_t i = 0; i < rank; ++i) {
                    const uint64_t b = row + lu[i];
                    c += (bits[b >> 5] >> (b & 31)) & 1u;
                }
                count += c;
            }
            uint64_t big = __ballot(bitmap && rank > 64);
            while (big) {
                const int src = __ffsll((unsigned long long)big) - 1;
                big &= big - 1;
                const uint32_t b_lo = __shfl(lo_u, src, kWave), b_rank = __shfl(rank, src, kWave);
                const uint64_t b_row = __shfl(row, src, kWave);
                const uint32_t *lu = dag_tgt + b_lo;
                uint32_t c = 0;
                for (uint32_t i = lane; i < b_rank; i += kWave) {
                    const uint64_t b = b_row + lu[i];
                    c += (bits[b >> 5] >> (b & 31)) & 1u;
                }
                count += c; // the total is a plain sum: any lane may carry any part of it
            }
            handled = handled || bitmap;
        }
        if (handled)
            continue;
        const uint32_t *lu = dag_tgt + loff[u];
        const uint32_t *lv = dag_tgt + loff[v];
        uint32_t nu = loff[u + 1] - loff[u], nv = loff[v + 1] - loff[v];
        const uint32_t *walk = lv, *probe = lu;
        uint32_t nwalk = nv, nprobe = nu;
        if (STRICT && nu < nv) { // both are sets: the intersection is symmetric, walk the shorter
            walk = lu;
            probe = lv;
            nwalk = nu;
            nprobe = nv;
        }
        for (uint32_t i = 0; i < nwalk; ++i)
            count += tc_contains(probe, nprobe, walk[i]) ? 1u : 0u;
    }
    const uint64_t block_total = block_sum<uint64_t, TC_WAVES>(count, red);
    if (threadIdx.x == 0 && block_total)
        atomicAdd(total, (unsigned long long)block_total);
}

} // namespace

GM_API int gm_triangle_count(const gm_csr *g, uint64_t *triangles_out)
{
    GM_CHECK(g && triangles_out, GM_ERR_INVALID, "gm_triangle_count: null argument");
    *triangles_out = 0;
    const uint32_t n = (uint32_t)g->n;
    if (n == 0 || g->m == 0)
        return GM_OK;
    gm::DeviceGuard guard(g->device);
    gm::DevBuf low_len, loff, ctrl;
    GM_TRY(low_len.alloc(((size_t)n + 1) * 4));
    GM_TRY(loff.alloc(((size_t)n + 1) * 4));
    GM_TRY(ctrl.alloc(16));
    GM_HIP(hipMemset(ctrl.p, 0, 16));
    unsigned grid = gm::div_up((uint64_t)n + 1, TC_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    {
        gm::DevBuf row_start;
        GM_TRY(row_start.alloc(((size_t)g->m / 32 + 1) * 4));
        GM_HIP(hipMemset(row_start.p, 0, row_start.bytes));
        hipLaunchKernelGGL(tc_low_len_kernel, dim3(grid), dim3(TC_BLOCK), 0, 0, g->offsets, g->targets, n,
                           low_len.as<uint32_t>(), row_start.as<uint32_t>(), ctrl.as<uint32_t>() + 2);
        unsigned egrid = gm::div_up(g->m, TC_BLOCK);
        if (egrid > 256 * 32)
            egrid = 256 * 32;
        hipLaunchKernelGGL(tc_order_kernel, dim3(egrid), dim3(TC_BLOCK), 0, 0, g->targets, g->m, row_start.as<uint32_t>(),
                           ctrl.as<uint32_t>() + 2);
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize()); // row_start is released on scope exit
    }
    {
        size_t tmp_bytes = 0;
        GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, low_len.as<uint32_t>(), loff.as<uint32_t>(), 0u,
                                       (size_t)n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        gm::DevBuf tmp;
        GM_TRY(tmp.alloc(tmp_bytes));
        GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, low_len.as<uint32_t>(), loff.as<uint32_t>(), 0u,
                                       (size_t)n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        GM_HIP(hipDeviceSynchronize());
    }
    uint32_t flags = 0, dag_m = 0;
    GM_HIP(hipMemcpy(&flags, ctrl.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost));
    GM_CHECK((flags & 1u) == 0, GM_ERR_UNSUPPORTED,
             "gm_triangle_count: neighbour lists are not sorted (use CsrLayout::Sorted or Deduplicated)");
    GM_HIP(hipMemcpy(&dag_m, loff.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
    if (dag_m == 0)
        return GM_OK;
    gm::DevBuf dag_src, dag_tgt;
    GM_TRY(dag_src.alloc((size_t)dag_m * 4));
    GM_TRY(dag_tgt.alloc((size_t)dag_m * 4));
    hipLaunchKernelGGL(tc_dag_kernel, dim3(grid), dim3(TC_BLOCK), 0, 0, g->offsets, g->targets, loff.as<uint32_t>(), n,
                       dag_src.as<uint32_t>(), dag_tgt.as<uint32_t>());
    unsigned cgrid = gm::div_up(dag_m, TC_BLOCK);
    if (cgrid > 256 * 16)
        cgrid = 256 * 16;
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(ctrl.p);
    if (flags & 2u) {
        hipLaunchKernelGGL(tc_count_kernel<false>, dim3(cgrid), dim3(TC_BLOCK), 0, 0, loff.as<uint32_t>(),
                           dag_src.as<uint32_t>(), dag_tgt.as<uint32_t>(), (uint64_t)dag_m, (const uint32_t *)nullptr, 0u,
                           d_total);
    } else {
        // strictly increasing lists: membership in the prefix lists of the K smallest ids through a
        // triangular bitmap (K = 262144 -> 4.3 GB of the 288 GB; GM_TC_K overrides, 0 disables)
        uint32_t K = n < (1u << 18) ? n : (1u << 18);
        if (const char *e = getenv("GM_TC_K"))
            K = (uint32_t)atoll(e) < n ? (uint32_t)atoll(e) : n;
        gm::DevBuf bits;
        // the bitmap is an accelerator, not a requirement: when HBM is short (a graph that already fills the
        // card, a shared node) halve K until it fits, down to the search-only path
        while (K >= 2) {
            const uint64_t nbits = (uint64_t)K * (K - 1) / 2;
            if (bits.alloc((size_t)((nbits + 31) / 32 + 1) * 4) == GM_OK)
                break;
            (void)hipGetLastError();
            K /= 2;
        }
        if (K >= 2) {
            const uint64_t nbits = (uint64_t)K * (K - 1) / 2;
            const size_t words = (size_t)((nbits + 31) / 32) + 1;
            GM_HIP(hipMemset(bits.p, 0, words * 4));
            hipLaunchKernelGGL(tc_bitmap_fill_kernel, dim3(cgrid), dim3(TC_BLOCK), 0, 0, dag_src.as<uint32_t>(),
                               dag_tgt.as<uint32_t>(), (uint64_t)dag_m, K, bits.as<uint32_t>());
        } else {
            K = 0;
        }
        hipLaunchKernelGGL(tc_count_kernel<true>, dim3(cgrid), dim3(TC_BLOCK), 0, 0, loff.as<uint32_t>(),
                           dag_src.as<uint32_t>(), dag_tgt.as<uint32_t>(), (uint64_t)dag_m,
                           K ? bits.as<uint32_t>() : (const uint32_t *)nullptr, K, d_total);
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize()); // bits is released on scope exit
    }
    GM_HIP(hipGetLastError());
    unsigned long long total = 0;
    GM_HIP(hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost));
    *triangles_out = total;
    return GM_OK;
}
