// wcc.hip — weakly connected components (Afforest) on device-resident out-/in-CSR.
//
// Replaces crates/algos/src/wcc.rs:103-301 + crates/algos/src/afforest.rs:22-56.
// The parent array lives in HBM (u32[n]); `link` is the reference's Afforest::union with
// atomicCAS (agent scope) and L1-bypassing loads — a plain load could spin forever on a stale L1
// line that another CU has since rewritten (MI355X_MICROARCH.md, inter-workgroup visibility).
// Invariant parent[x] <= x  =>  after the final compress parent[u] is the minimum node id of u's
// component, independent of schedule: results are bit-exact with the reference's
// Components::component(u) for wcc_afforest, wcc_afforest_dss and wcc_baseline alike.
//
// Pipeline (wcc.rs:164-182): sample_subgraph -> compress -> find_largest_component ->
// link_remaining -> compress.  Work distribution: one lane per node for short lists; lists longer
// than 32 entries are spread over the node's whole wavefront (RMAT hubs).
#include "common.hpp"
#include "device_utils.hpp"

namespace {

using namespace gm;

constexpr int WCC_BLOCK = 256;
constexpr uint32_t WCC_COOP = 32; // lists longer than this are linked by the whole wavefront

// afforest.rs:22-39, from any node p1 of u's tree and any node p2 of v's (the parents the caller has loaded: several
// independent loads in flight instead of one after the other; a parent read a moment ago is still a node of the same tree —
// trees only merge — and the loop climbs from wherever it starts)
__device__ __forceinline__ void af_link_from(uint32_t *parent, uint32_t p1, uint32_t p2)
{
    while (p1 != p2) {
        const uint32_t high = p1 > p2 ? p1 : p2;
        const uint32_t low = p1 + p2 - high;
        const uint32_t p_high = ld_agent(&parent[high]);
        if (p_high == low)
            break;
        if (p_high == high && atomicCAS(&parent[high], high, low) == high)
            break;
        p1 = ld_agent(&parent[ld_agent(&parent[high])]);
        p2 = ld_agent(&parent[low]);
    }
}

__device__ __forceinline__ void af_link(uint32_t *parent, uint32_t u, uint32_t v)
{
    const uint32_t p1 = ld_agent(&parent[u]);
    const uint32_t p2 = ld_agent(&parent[v]);
    af_link_from(parent, p1, p2);
}

__global__ void wcc_init_kernel(uint32_t *__restrict__ parent, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride)
        parent[u] = u;
}

// afforest.rs:50-56
__global__ void wcc_compress_kernel(uint32_t *parent, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) {
        uint32_t p = ld_agent(&parent[u]);
        uint32_t pp = ld_agent(&parent[p]);
        while (p != pp) {
            st_agent(&parent[u], pp);
            p = pp;
            pp = ld_agent(&parent[p]);
        }
    }
}

// wcc.rs:186-204: link u with its first `rounds` out-neighbours
// (Round 6 measured two ways of putting more of a node's chain of dependent accesses in flight — 4 or 8 nodes per lane with
// every level of the chain loaded for all of them before the next, and unions of two known roots going straight to their CAS:
// 257 -> 257 / 330 us at scale 22, tools/runs/r06_call05.sh / r06_call06.sh.  The kernel is bound by the number of random
// 128-byte transactions and device-scope atomics, not by their latency; a CAS that fails costs more than the load it replaced.)
__global__ void wcc_sample_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt,
                                  uint32_t *parent, uint32_t n, uint64_t rounds)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) {
        const uint32_t s = off[u], e = off[u + 1];
        const uint64_t take = (uint64_t)(e - s) < rounds ? (uint64_t)(e - s) : rounds;
        for (uint64_t k = 0; k < take; ++k)
            af_link(parent, u, tgt[s + k]);
    }
}

// Hub lists are not linked by the wavefront that met them (a 300k-entry list would be the kernel's
// critical path: measured 6.6 ms of a 7 ms pass at scale 22) but cut into WCC_CHUNK-entry pieces for
// wcc_chunk_kernel, which spreads them over the whole grid.  The order of unions never matters.
constexpr uint32_t WCC_BIG = 2048, WCC_CHUNK = 1024;
struct WccChunks {
    uint4 *items = nullptr;   // {u, first entry, end of the list, 0 = out-list / 1 = in-list}
    uint32_t *count = nullptr; // null: no deferral (hubs are linked in place)
};

// Link u with list[s..e): short lists by the owning lane, long lists by the whole wavefront.
// Must be called by all 64 lanes of a wavefront (lanes without work pass s == e).
__device__ __forceinline__ void wcc_link_list(const uint32_t *__restrict__ tgt, uint32_t *parent, uint32_t u,
                                              uint32_t s, uint32_t e, WccChunks chunks, uint32_t which)
{
    const uint32_t len = e - s;
    if (len <= WCC_COOP)
        for (uint32_t i = s; i < e; ++i)
            af_link(parent, u, tgt[i]);
    const bool defer = chunks.count != nullptr && len > WCC_BIG;
    uint64_t big = __ballot(len > WCC_COOP && !defer);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    while (big) {
        const int src = __ffsll((unsigned long long)big) - 1;
        big &= big - 1;
        const uint32_t bu = __shfl(u, src, kWave), bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
        for (uint32_t i = bs + lane; i < be; i += kWave)
            af_link(parent, bu, tgt[i]);
    }
    uint64_t huge = __ballot(defer);
    while (huge) {
        const int src = __ffsll((unsigned long long)huge) - 1;
        huge &= huge - 1;
        const uint32_t hu = __shfl(u, src, kWave), hs = __shfl(s, src, kWave), he = __shfl(e, src, kWave);
        const uint32_t nch = (he - hs + WCC_CHUNK - 1u) / WCC_CHUNK;
        uint32_t first = 0;
        if (lane == 0)
            first = atomicAdd(chunks.count, nch);
        first = __shfl(first, 0, kWave);
        for (uint32_t c = lane; c < nch; c += kWave)
            chunks.items[first + c] = make_uint4(hu, hs + c * WCC_CHUNK, he, which);
    }
}

__global__ __launch_bounds__(WCC_BLOCK) void wcc_chunk_kernel(const uint32_t *__restrict__ out_tgt,
                                                              const uint32_t *__restrict__ in_tgt, uint32_t *parent,
                                                              const uint4 *__restrict__ items, const uint32_t *count)
{
    const uint32_t nchunks = *count;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const uint4 ch = items[c];
        const uint32_t *tgt = ch.w ? in_tgt : out_tgt;
        const uint32_t end = ch.y + WCC_CHUNK < ch.z ? ch.y + WCC_CHUNK : ch.z;
        for (uint32_t i = ch.y + lane; i < end; i += kWave)
            af_link(parent, ch.x, tgt[i]);
    }
}

// wcc.rs:274-301 (skip = UINT32_MAX never matches a parent: link everything, used by wcc_baseline
// with rounds = 0 and no in-CSR)
__global__ __launch_bounds__(WCC_BLOCK) void wcc_link_remaining_kernel(
    const uint32_t *__restrict__ out_off, const uint32_t *__restrict__ out_tgt, const uint32_t *__restrict__ in_off,
    const uint32_t *__restrict__ in_tgt, uint32_t *parent, uint32_t n, uint64_t rounds, const uint32_t *skip_ptr,
    uint32_t row_base /* node id of CSR row 0 (row slices of a partitioned graph) */, WccChunks chunks)
{
    const uint32_t skip = skip_ptr ? *skip_ptr : 0xFFFFFFFFu;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + kWave - 1) / kWave * kWave; // whole wavefronts enter the loop together
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_pad; r += stride) {
        const uint32_t u = row_base + r;
        const bool active = r < n && ld_agent(&parent[u]) != skip;
        uint32_t s = 0, e = 0;
        if (active) {
            s = out_off[r];
            e = out_off[r + 1];
            s = (uint64_t)(e - s) > rounds ? s + (uint32_t)rounds : e;
        }
        wcc_link_list(out_tgt, parent, u, s, e, chunks, 0u);
        if (in_off) {
            s = e = 0;
            if (active) {
                s = in_off[r];
                e = in_off[r + 1];
            }
            wcc_link_list(in_tgt, parent, u, s, e, chunks, 1u);
        }
    }
}

constexpr uint32_t WCC_MODE_LDS = 4096;

// wcc.rs:245-271: most frequent component among `samples` random nodes.  The reference draws from
// an unseeded WyRand; the choice only decides which component link_remaining skips and never
// changes the result.  One workgroup; ties -> smallest id.
// Up to WCC_MODE_LDS / 2 samples (the default is 1024) are COUNTED in an LDS hash table — one insertion per sample;
// until round 6 every sample was compared with every other (samples^2 LDS reads: 51 us of a 418 us pipeline at scale 22).
// More samples than that: the quadratic count, from LDS or from the work buffer.
__global__ __launch_bounds__(WCC_BLOCK) void wcc_sample_mode_kernel(const uint32_t *parent, uint32_t n,
                                                                     uint32_t samples, uint64_t seed,
                                                                     uint32_t *__restrict__ sample_buf,
                                                                     uint32_t *__restrict__ skip_out)
{
    __shared__ unsigned long long best; // (count << 32) | ~id  -> max picks highest count, then smallest id
    __shared__ uint32_t staged[WCC_MODE_LDS]; // the samples (quadratic count) / the table's keys (hash count)
    __shared__ uint32_t counts[WCC_MODE_LDS];
    if (threadIdx.x == 0)
        best = 0ull;
    const bool hashed = samples <= WCC_MODE_LDS / 2;
    const bool in_lds = samples <= WCC_MODE_LDS;
    if (hashed)
        for (uint32_t i = threadIdx.x; i < WCC_MODE_LDS; i += WCC_BLOCK) {
            staged[i] = 0xFFFFFFFFu; // (no label: labels are node ids, at most 2^32 - 2)
            counts[i] = 0u;
        }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < samples; k += WCC_BLOCK) {
        uint64_t x = seed + k;
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        const uint32_t c = ld_agent(&parent[(uint32_t)(x % n)]);
        if (hashed) {
            uint32_t h = (c * 2654435761u) >> 20; // 12 bits
            for (;;) {
                const uint32_t prev = atomicCAS(&staged[h], 0xFFFFFFFFu, c);
                if (prev == 0xFFFFFFFFu || prev == c) {
                    atomicAdd(&counts[h], 1u);
                    break;
                }
                h = (h + 1u) & (WCC_MODE_LDS - 1u); // (at most samples <= half the table's keys: a free slot exists)
            }
        } else if (in_lds)
            staged[k] = c;
        else
            sample_buf[k] = c;
    }
    __syncthreads();
    if (hashed) {
        for (uint32_t i = threadIdx.x; i < WCC_MODE_LDS; i += WCC_BLOCK)
            if (staged[i] != 0xFFFFFFFFu)
                atomicMax(&best, ((unsigned long long)counts[i] << 32) | (uint32_t)~staged[i]);
    } else {
        for (uint32_t k = threadIdx.x; k < samples; k += WCC_BLOCK) {
            uint32_t cnt = 0;
            uint32_t c;
            if (in_lds) { // every lane reads the same word per step: an LDS broadcast
                c = staged[k];
                for (uint32_t j = 0; j < samples; ++j)
                    cnt += staged[j] == c;
            } else {
                c = sample_buf[k];
                for (uint32_t j = 0; j < samples; ++j)
                    cnt += sample_buf[j] == c;
            }
            atomicMax(&best, ((unsigned long long)cnt << 32) | (uint32_t)~c);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
        *skip_out = ~(uint32_t)best;
}

unsigned wcc_grid(uint64_t n)
{
    unsigned g = gm::div_up(n, WCC_BLOCK);
    return g > 256 * 8 ? 256 * 8 : (g ? g : 1);
}

} // namespace

namespace gm {
// The buffers of a call, taken from the out-CSR's handle and parked there again on every way out (hipMalloc /
// hipFree of the two were a quarter of a scale-22 call).
struct WccScratchLease {
    const gm_csr *g;
    std::unique_ptr<WccScratch> sc;
    explicit WccScratchLease(const gm_csr *csr) : g(csr)
    {
        {
            std::lock_guard<std::mutex> lock(g->cache_mu);
            sc = std::move(g->wcc_scratch);
        }
        if (!sc)
            sc.reset(new (std::nothrow) WccScratch);
    }
    ~WccScratchLease()
    {
        std::lock_guard<std::mutex> lock(g->cache_mu);
        if (sc && !g->wcc_scratch)
            g->wcc_scratch = std::move(sc);
    }
};

// Shared by gm_wcc_afforest / gm_wcc_baseline; leaves the labels in d_parent (u32[n]).
int wcc_device(const gm_csr *out_csr, const gm_csr *in_csr, uint64_t rounds, uint64_t sampling, bool afforest,
               uint32_t *d_parent, hipStream_t st, DevBuf &buf /* work space, grown when too small */)
{
    const uint32_t n = (uint32_t)out_csr->n;
    const unsigned grid = wcc_grid(n);
    // deferred hub chunks: at most one per WCC_CHUNK entries plus one per hub list, out- and in-lists together
    const uint64_t entries = out_csr->m + (in_csr ? in_csr->m : 0);
    // one allocation: [chunk count | pad] [chunk items] [sample buffer + skip id]
    const size_t items_bytes = (size_t)(entries / WCC_CHUNK + entries / WCC_BIG + 64) * sizeof(uint4);
    const size_t need = 16 + items_bytes + (size_t)(sampling + 1) * 4;
    if (buf.bytes < need)
        GM_TRY(buf.alloc(need));
    GM_HIP(hipMemsetAsync(buf.p, 0, 16, st));
    const WccChunks chunks{reinterpret_cast<uint4 *>(buf.as<char>() + 16), buf.as<uint32_t>()};
    uint32_t *sample_buf = reinterpret_cast<uint32_t *>(buf.as<char>() + 16 + items_bytes);
    hipLaunchKernelGGL(wcc_init_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, d_parent, n);
    if (!afforest) { // wcc.rs:103-122: union every out-edge
        hipLaunchKernelGGL(wcc_link_remaining_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, out_csr->offsets,
                           out_csr->targets, (const uint32_t *)nullptr, (const uint32_t *)nullptr, d_parent, n,
                           (uint64_t)0, (const uint32_t *)nullptr, 0u, chunks);
        hipLaunchKernelGGL(wcc_chunk_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, out_csr->targets,
                           (const uint32_t *)nullptr, d_parent, chunks.items, chunks.count);
        hipLaunchKernelGGL(wcc_compress_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, d_parent, n);
        GM_HIP(hipGetLastError());
        GM_HIP(hipStreamSynchronize(st)); // the work buffer is freed on return
        return GM_OK;
    }
    uint32_t *d_skip = sample_buf + sampling;
    gm::PhaseTimer timer(st); // phase names as logged by the reference, wcc.rs:164-182
    hipLaunchKernelGGL(wcc_sample_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, out_csr->offsets, out_csr->targets,
                       d_parent, n, rounds);
    timer.done("Link subgraph");
    hipLaunchKernelGGL(wcc_compress_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, d_parent, n);
    timer.done("Sample compress");
    hipLaunchKernelGGL(wcc_sample_mode_kernel, dim3(1), dim3(WCC_BLOCK), 0, st, d_parent, n, (uint32_t)sampling,
                       (uint64_t)0x2545F4914F6CDD1Dull, sample_buf, d_skip);
    timer.done("Get component");
    hipLaunchKernelGGL(wcc_link_remaining_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, out_csr->offsets,
                       out_csr->targets, in_csr->offsets, in_csr->targets, d_parent, n, rounds,
                       (const uint32_t *)d_skip, 0u, chunks);
    hipLaunchKernelGGL(wcc_chunk_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, out_csr->targets, in_csr->targets, d_parent,
                       chunks.items, chunks.count);
    timer.done("Link remaining");
    hipLaunchKernelGGL(wcc_compress_kernel, dim3(grid), dim3(WCC_BLOCK), 0, st, d_parent, n);
    timer.done("Final compress");
    GM_HIP(hipGetLastError());
    GM_HIP(hipStreamSynchronize(st)); // the work buffer is freed on return
    return GM_OK;
}
} // namespace gm

GM_API int gm_wcc_afforest(const gm_csr *out_csr, const gm_csr *in_csr, uint64_t neighbor_rounds,
                           uint64_t sampling_size, uint32_t *components_out)
{
    GM_CHECK(out_csr && in_csr, GM_ERR_INVALID, "gm_wcc_afforest: null CSR");
    GM_CHECK(out_csr->n == in_csr->n && out_csr->device == in_csr->device, GM_ERR_INVALID,
             "gm_wcc_afforest: out/in CSR disagree (n %llu vs %llu)", (unsigned long long)out_csr->n,
             (unsigned long long)in_csr->n);
    // wcc.rs:260-263: max_by(...).unwrap() on an empty sample map panics
    GM_CHECK(sampling_size > 0, GM_ERR_INVALID, "gm_wcc_afforest: sampling_size must be > 0 (reference panics)");
    GM_CHECK(sampling_size <= (1u << 20), GM_ERR_RANGE, "gm_wcc_afforest: sampling_size %llu too large",
             (unsigned long long)sampling_size);
    const uint64_t n = out_csr->n;
    // wcc.rs:256: generate_range(0..0) on an empty graph panics
    GM_CHECK(n > 0, GM_ERR_INVALID, "gm_wcc_afforest: empty graph (reference panics when sampling)");
    GM_CHECK(components_out, GM_ERR_INVALID, "gm_wcc_afforest: components_out is null");
    gm::DeviceGuard guard(out_csr->device);
    gm::WccScratchLease lease(out_csr);
    GM_CHECK(lease.sc, GM_ERR_NOMEM, "gm_wcc_afforest: out of host memory");
    gm::DevBuf &parent = lease.sc->labels;
    if (parent.bytes < n * 4)
        GM_TRY(parent.alloc(n * 4));
    GM_TRY(gm::wcc_device(out_csr, in_csr, neighbor_rounds, sampling_size, true, parent.as<uint32_t>(), 0, lease.sc->work));
    GM_HIP(hipMemcpy(components_out, parent.p, n * 4, hipMemcpyDefault)); // host (page-locked: link speed) or device memory
    return GM_OK;
}

GM_API int gm_wcc_baseline(const gm_csr *out_csr, uint32_t *components_out)
{
    GM_CHECK(out_csr, GM_ERR_INVALID, "gm_wcc_baseline: null CSR");
    const uint64_t n = out_csr->n;
    if (n == 0)
        return GM_OK;
    GM_CHECK(components_out, GM_ERR_INVALID, "gm_wcc_baseline: components_out is null");
    gm::DeviceGuard guard(out_csr->device);
    gm::WccScratchLease lease(out_csr);
    GM_CHECK(lease.sc, GM_ERR_NOMEM, "gm_wcc_baseline: out of host memory");
    gm::DevBuf &parent = lease.sc->labels;
    if (parent.bytes < n * 4)
        GM_TRY(parent.alloc(n * 4));
    GM_TRY(gm::wcc_device(out_csr, nullptr, 0, 0, false, parent.as<uint32_t>(), 0, lease.sc->work));
    GM_HIP(hipMemcpy(components_out, parent.p, n * 4, hipMemcpyDefault));
    return GM_OK;
}


// ------------------------------------------------------------------------------------------------
// Building blocks of the partitioned run (labels replicated, every rank links the edges
// of its own rows, min-all-reduce of the label vector between rounds until nothing changes).
// ------------------------------------------------------------------------------------------------
GM_API int gm_wcc_init_labels(uint64_t n, uint64_t d_parent, int device, void *stream)
{
    GM_CHECK(d_parent || n == 0, GM_ERR_INVALID, "gm_wcc_init_labels: null labels");
    GM_CHECK(n < (1ull << 32), GM_ERR_RANGE, "gm_wcc_init_labels: n exceeds u32");
    if (n == 0)
        return GM_OK;
    gm::DeviceGuard guard(device);
    hipLaunchKernelGGL(wcc_init_kernel, dim3(wcc_grid(n)), dim3(WCC_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<uint32_t *>(d_parent), (uint32_t)n);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_wcc_link_rows(const gm_csr *out_rows, const gm_csr *in_rows, uint64_t row_begin, uint64_t n_global,
                            uint64_t d_parent, void *stream)
{
    GM_CHECK(out_rows && d_parent, GM_ERR_INVALID, "gm_wcc_link_rows: null argument");
    GM_CHECK(!in_rows || in_rows->n == out_rows->n, GM_ERR_INVALID, "gm_wcc_link_rows: out/in slices disagree");
    GM_CHECK(row_begin + out_rows->n <= n_global && n_global < (1ull << 32), GM_ERR_RANGE,
             "gm_wcc_link_rows: rows [%llu, %llu) outside a graph of %llu nodes", (unsigned long long)row_begin,
             (unsigned long long)(row_begin + out_rows->n), (unsigned long long)n_global);
    gm::DeviceGuard guard(out_rows->device);
    uint32_t *parent = reinterpret_cast<uint32_t *>(d_parent);
    hipStream_t st = (hipStream_t)stream;
    if (out_rows->n)
        hipLaunchKernelGGL(wcc_link_remaining_kernel, dim3(wcc_grid(out_rows->n)), dim3(WCC_BLOCK), 0, st,
                           out_rows->offsets, out_rows->targets, in_rows ? in_rows->offsets : (const uint32_t *)nullptr,
                           in_rows ? in_rows->targets : (const uint32_t *)nullptr, parent, (uint32_t)out_rows->n,
                           (uint64_t)0, (const uint32_t *)nullptr, (uint32_t)row_begin, WccChunks{});
    hipLaunchKernelGGL(wcc_compress_kernel, dim3(wcc_grid(n_global)), dim3(WCC_BLOCK), 0, st, parent, (uint32_t)n_global);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

namespace gm {
void warm_wcc() // (common.hpp: the code object of this file, loaded ahead of an algorithm's first call)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&wcc_init_kernel)) != hipSuccess)
        (void)hipGetLastError();
}
} // namespace gm
