// multi.hip — page_rank() 1-D partitioned over the GPUs of one node, behind the C ABI.
//
// BASELINE north_star: "Graphs larger than one GPU are 1-D vertex-range partitioned across the 8 MI355X of one
// node with a RCCL all-reduce of the rank vector over xGMI at each iteration" — reachable from the host language
// of the reference (Rust, over the C ABI) without Python: one host thread drives every device (hipSetDevice +
// one HIP stream per device), RCCL is initialised with ncclCommInitAll and the collectives of a sweep are issued
// inside one ncclGroupStart / ncclGroupEnd.
//
//   partition   contiguous row ranges balanced by in-degree: the reference's greedy walk, in_degree_partition +
//               greedy_node_map_partition, crates/builder/src/graph_ops.rs:431-439,479-509
//   exchange    only nodes WITH out-edges are ever gathered (the out_score of the others is +inf and never read,
//               page_rank.rs:78,158): they are numbered rank-major (ascending node ids: the order in which the reference
//               adds a row's terms, which the engines' hub rows follow) with a fixed per-rank stride, every rank's
//               targets are rewritten into that index space once, and the all-gather output is consumed directly
//               as the next sweep's x vector (an all-gather moves half the bytes of the all-reduce of a
//               zero-padded vector and gives the same result)
//   per sweep   the exchanged vector is cut into K regions (GM_MULTI_PARTS, default 2; region k = row group k of every
//               rank, whole source tiles) and every rank's rows into K groups: propagate region 0 as soon as it has
//               landed, then region 1; accumulate group 0 -> compact it -> START its all-gather on the rank's exchange
//               stream -> accumulate group 1 -> start its all-gather: region k travels under the work on the other
//               groups and under the next sweep's propagation of the regions before it.  Events order the two
//               streams; the host synchronises only to read the error (never, when tolerance == 0).  Slices too
//               small for propagation-blocking engines run one whole sweep + one all-gather per sweep.
//   residency   partition, slices, engines, buffers, streams, events and communicators are parked in the in-CSR's
//               handle (gm::MultiState): a second call on the same graph / device list only re-initialises the scores
//
// Rows below the hub threshold get the same bits as the single-GPU run (exactly rounded row sums do not depend on
// the partition); hub rows agree to ~1e-6 (their step boundaries move with the bin layout).
//
// `devices` may name one GPU several times ("virtual ranks": the partition, the index rewrite and the sweep
// drivers exercised on a single GPU); the exchange is then done with device-to-device copies, RCCL refuses
// duplicate devices.  librccl.so is loaded on first use (dlopen): the single-GPU library has no RCCL dependency.
#include "common.hpp"
#include "device_utils.hpp"

#include <rccl/rccl.h>
#include <rocprim/rocprim.hpp>

#include <dlfcn.h>

#include <algorithm>
#include <memory>
#include <vector>

extern char **environ; // unistd.h's, declared here at global scope

namespace {

using namespace gm;

// ---- RCCL, bound at run time ---------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int rccl_get(const Rccl **out)
{
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.lib) {
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        GM_CHECK(h, GM_ERR_UNSUPPORTED, "gm_page_rank_multi: librccl.so not found (%s)", dlerror());
        Rccl t;
        t.lib = h;
        t.CommInitAll = reinterpret_cast<decltype(t.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        t.CommDestroy = reinterpret_cast<decltype(t.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        t.AllGather = reinterpret_cast<decltype(t.AllGather)>(dlsym(h, "ncclAllGather"));
        t.Broadcast = reinterpret_cast<decltype(t.Broadcast)>(dlsym(h, "ncclBroadcast"));
        t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(dlsym(h, "ncclGroupStart"));
        t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        GM_CHECK(t.CommInitAll && t.CommDestroy && t.AllGather && t.Broadcast && t.GroupStart && t.GroupEnd && t.GetErrorString,
                 GM_ERR_UNSUPPORTED, "gm_page_rank_multi: librccl.so lacks an expected symbol");
        r = t;
    }
    *out = &r;
    return GM_OK;
}

#define GM_NCCL(rc, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t gm_n_ = (expr);                                                                          \
        if (gm_n_ != ncclSuccess) {                                                                           \
            gm::set_error("%s failed: %s", #expr, (rc)->GetErrorString(gm_n_));                               \
            return GM_ERR_HIP;                                                                                \
        }                                                                                                     \
    } while (0)

// ---- kernels ----------------------------------------------------------------------------------------------
__global__ void mg_has_out_degree_kernel(const uint32_t *__restrict__ outdeg, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u <= n; u += stride)
        flag[u] = (u < n && outdeg[u] != 0u) ? 1u : 0u;
}

// local rows (of [lo, hi)) that have out-edges, in order: what the rank contributes to the exchange
__global__ void mg_send_rows_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t lo,
                                    uint32_t hi, uint32_t rank_lo, uint32_t *__restrict__ rows)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t v = lo + blockIdx.x * blockDim.x + threadIdx.x; v < hi; v += s)
        if (flag[v])
            rows[pos[v] - pos[lo]] = v - rank_lo;
}

__global__ void mg_out_degree_kernel(const uint32_t *__restrict__ out_off, uint32_t lo, uint32_t count,
                                     uint32_t *__restrict__ deg)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += s)
        deg[i] = out_off[lo + i + 1] - out_off[lo + i];
}

__global__ void mg_rebase_kernel(uint32_t *__restrict__ off, uint32_t count, uint32_t first)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += s)
        off[i] -= first;
}

__global__ void mg_map_targets_kernel(uint32_t *__restrict__ tgt, uint64_t count, const uint32_t *__restrict__ node_map)
{
    const uint64_t s = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += s)
        tgt[i] = node_map[tgt[i]];
}

__global__ void mg_compact_kernel(const float *__restrict__ x_loc, const uint32_t *__restrict__ rows, uint32_t count,
                                  float *__restrict__ x_send)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += s)
        x_send[i] = x_loc[rows[i]];
}

// device-to-device copy between (possibly equal) devices
hipError_t mg_copy(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes)
{
    if (bytes == 0)
        return hipSuccess;
    return dst_dev == src_dev ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, dst_dev, src, src_dev, bytes);
}

unsigned mg_grid(uint64_t count)
{
    const uint64_t g = (count + 255) / 256;
    return (unsigned)(g > 8192 ? 8192 : (g ? g : 1));
}

// graph_ops.rs:479-509 (greedy_node_map_partition over in-degrees, :431-439): walk the nodes in order, close a
// range as soon as its in-degree sum reaches ceil(edge_count / concurrency) while fewer than concurrency - 1
// ranges exist; the last range ends at node_count.  Returns concurrency + 1 bounds (trailing ranks may be empty).
std::vector<uint32_t> greedy_in_degree_bounds(const std::vector<uint32_t> &off, uint32_t parts)
{
    const uint32_t n = (uint32_t)off.size() - 1;
    std::vector<uint32_t> bounds{0u};
    const uint64_t total = off[n], batch = total ? (total + parts - 1) / parts : 0;
    uint32_t start = 0;
    while (start < n && bounds.size() < parts) {
        // first node u >= start with off[u + 1] - off[start] >= batch
        const uint64_t want = (uint64_t)off[start] + batch;
        uint32_t u = (uint32_t)(std::lower_bound(off.begin() + start + 1, off.end(), want,
                                                  [](uint32_t a, uint64_t b) { return (uint64_t)a < b; }) -
                                 off.begin());
        u = u ? u - 1 : 0; // index into off of the first prefix >= want, minus one = the node
        if (u < start)
            u = start;
        if (u >= n - 1)
            break;
        bounds.push_back(u + 1);
        start = u + 1;
    }
    while (bounds.size() < (size_t)parts + 1)
        bounds.push_back(n);
    return bounds;
}

constexpr uint32_t MG_ROW_ALIGN = 16384;   // row splits of a sweep in pieces: a multiple of any plan's rows per bin
constexpr uint32_t MG_SOURCE_TILE = 32768; // regions of the exchanged vector: a multiple of any plan's source tile
constexpr uint32_t MG_MAX_PARTS = 4;

struct Rank {
    int device = 0;
    uint32_t lo = 0, hi = 0;
    uint32_t send_count[MG_MAX_PARTS] = {0, 0, 0, 0};
    hipStream_t st = nullptr;  // the sweep kernels
    hipStream_t cst = nullptr; // the exchange: runs under the kernels of the other row groups
    hipEvent_t ev_ready[MG_MAX_PARTS] = {};   // x_send[k] compacted (recorded on st)
    hipEvent_t ev_done[2][MG_MAX_PARTS] = {}; // region k of x[buf] complete (recorded on cst)
    gm_csr *rows = nullptr; // row slice with targets in exchange index space (lives on `device`)
    gm_pr *pr = nullptr;
    DevBuf outdeg, scores, x_loc, x[2], x_send[MG_MAX_PARTS], send_rows[MG_MAX_PARTS], err;
    ~Rank()
    {
        DeviceGuard g(device);
        if (st)
            (void)hipStreamSynchronize(st);
        if (cst)
            (void)hipStreamSynchronize(cst);
        if (pr)
            gm_pr_destroy(pr);
        if (rows)
            gm_csr_free(rows);
        // the buffers go while the rank's device is current (members are destroyed after this body, when the guard has
        // already restored the caller's device)
        outdeg.release(), scores.release(), x_loc.release(), x[0].release(), x[1].release(), err.release();
        for (uint32_t k = 0; k < MG_MAX_PARTS; ++k)
            x_send[k].release(), send_rows[k].release();
        for (uint32_t k = 0; k < MG_MAX_PARTS; ++k) {
            if (ev_ready[k])
                (void)hipEventDestroy(ev_ready[k]);
            for (int b = 0; b < 2; ++b)
                if (ev_done[b][k])
                    (void)hipEventDestroy(ev_done[b][k]);
        }
        if (st)
            (void)hipStreamDestroy(st);
        if (cst)
            (void)hipStreamDestroy(cst);
    }
};

struct Comms {
    const Rccl *rc = nullptr;
    std::vector<ncclComm_t> c;
    ~Comms()
    {
        for (ncclComm_t k : c)
            if (k && rc)
                (void)rc->CommDestroy(k);
    }
};

// node v of rank p, row group k (splits[p * (K + 1) + k] <= v < ... + k + 1], global ids) with out-edges -> slot
// p * rank_stride + group_off[k] + (its rank among the group's nodes with out-edges); others are never gathered.
// RANK-MAJOR: the slots ascend with the node ids, so a row's terms reach the engines in the order the reference adds them
__global__ void mg_node_map_parts_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                         const uint32_t *__restrict__ splits, uint32_t P, uint32_t K,
                                         uint32_t rank_stride, const uint32_t *__restrict__ region_off, uint32_t n,
                                         uint32_t *__restrict__ node_map)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += s) {
        uint32_t p = 0; // P <= 64, K <= 4: short scans
        while (p + 1 < P && v >= splits[(p + 1) * (K + 1)])
            ++p;
        uint32_t k = 0;
        while (k + 1 < K && v >= splits[p * (K + 1) + k + 1])
            ++k;
        node_map[v] = flag[v] ? p * rank_stride + region_off[k] + (pos[v] - pos[splits[p * (K + 1) + k]]) : 0xFFFFFFFFu;
    }
}

} // namespace

// Everything a partitioned run derives from (out CSR, in CSR, device list, damping): the partition, the exchange
// layout, every rank's row slice / engine / buffers / streams / events, the communicators.  Parked in the in-CSR's
// handle between calls (the reference's app calls page_rank in a loop): a second call only re-initialises the scores.
struct gm::MultiState {
    const gm_csr *out_csr = nullptr;
    std::vector<int> devs;
    float damping = 0.f;
    int engine = 0;
    uint64_t env_hash = 0; // the plan- and layout-shaping GM_PB_* / GM_MULTI_* knobs the engines were built under (multi_env_hash: by name)
    uint32_t K = 1;
    uint32_t P = 0, n = 0;
    bool distinct = true, pieces = false;
    std::vector<uint32_t> bounds, strides, region_off; // region_off[k]: where row group k starts inside a rank's stretch of x
    uint32_t rank_stride = 0;                          // floats per rank in x (sum of the strides)
    uint64_t x_len = 0;
    std::vector<std::unique_ptr<Rank>> ranks;
    Comms comms;
    PinnedBuf herr;
    ~MultiState()
    {
        ranks.clear(); // streams drained and destroyed before the communicators go
    }
};

void gm::MultiStateDeleter::operator()(gm::MultiState *p) const { delete p; }

namespace {

using MultiPtr = std::unique_ptr<gm::MultiState, gm::MultiStateDeleter>;

// What a rank is built from, all of it on the rank's own device: its rows of the in-CSR (offsets rebased to 0, targets as
// GLOBAL node ids) and the out-degree of EVERY node (which nodes are ever gathered, and its own rows' divisors).  Consumed.
// which local rows have MORE THAN ONE in-edge (one byte per row): the others' scores are constants or copies of one node's
__global__ void mg_has_in_kernel(const uint32_t *__restrict__ off, uint32_t rows, uint8_t *__restrict__ out)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride)
        out[r] = off[r + 1] - off[r] > 1u ? 1 : 0;
}

// per entry of the exchange vector: does the node in that slot have at most one in-edge?  (gm_csr_set_source_flags)
__global__ void mg_slot_flags_kernel(const uint32_t *__restrict__ node_map, const uint8_t *__restrict__ has_in, uint32_t n,
                                     uint8_t *__restrict__ slot_flags)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride)
        if (node_map[v] != 0xFFFFFFFFu && !has_in[v])
            slot_flags[node_map[v]] = 1;
}

struct RankInput {
    int device = 0;
    DevBuf off, tgt, outdeg_full;
    uint64_t edges = 0;
};

// the plan / engine knobs of the environment as one number: a parked state built under other knobs is not reused
__global__ void mg_max_target_kernel(const uint32_t *__restrict__ tgt, uint64_t m, uint32_t *__restrict__ out)
{
    uint32_t mx = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        mx = tgt[i] > mx ? tgt[i] : mx;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t w = __shfl_xor(mx, o, kWave);
        mx = w > mx ? w : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && mx)
        atomicMax(out, mx);
}

uint64_t multi_env_hash()
{
    // the knobs that shape a plan or the run's layout, by name (getenv, like everywhere else; the *_NOCACHE switches and the
    // per-launch measurement knobs are not among them: toggling those must not look like another configuration)
    static const char *const knobs[] = {"GM_PB_HOT",       "GM_PB_RB",        "GM_PB_SLOG",      "GM_PB_CHUNK",     "GM_PB_SPLIT",
                                        "GM_PB_WGS",       "GM_PB_COMPACT",   "GM_PB_ORDER",     "GM_PB_XCD",       "GM_PB_HUB_DEG",
                                        "GM_PB_HUB_GROUP", "GM_PB_HUB_LONG",  "GM_PB_HUB_HOT",   "GM_PB_HUB_CSR",   "GM_PB_HUB_ROOM", "GM_PB_HUB_THIN", "GM_PB_HOT_TRIM",
                                        "GM_PB_HUB_FORK",  "GM_PB_LONG_PASSES", "GM_PB_TIERS",   "GM_PB_HOT16",     "GM_PB_SEGPAD",
                                        "GM_PB_SPREAD",    "GM_PB_SPREAD_PLAN", "GM_PB_FILTER_BITS", "GM_PB_WG_GROUP", "GM_PB_BIN_GAP",
                                        "GM_MULTI_ENGINE", "GM_MULTI_PARTS",  "GM_PB_HUB_LEAVES", "GM_PB_HUB_REGROUP", "GM_PB_HUB_SLICE_GROUPS"};
    uint64_t h = 1469598103934665603ull;
    for (const char *name : knobs) {
        const char *v = getenv(name);
        for (const char *c = name; *c; ++c)
            h = (h ^ (uint64_t)(unsigned char)*c) * 1099511628211ull;
        h = (h ^ 0x3Dull) * 1099511628211ull;
        for (const char *c = v ? v : ""; *c; ++c)
            h = (h ^ (uint64_t)(unsigned char)*c) * 1099511628211ull;
    }
    return h;
}

// The partitioned run's state from per-rank inputs: nothing here touches a device that holds "the whole graph" — every
// rank derives the exchange layout from the out-degree vector on its own device (a scan of n flags: the same numbers on
// every rank) and rewrites its own targets.
int multi_build_from(std::vector<RankInput> &in, const std::vector<uint32_t> &bounds_in, uint32_t n, const std::vector<int> &devs,
                     bool distinct, float damping_factor, int engine_env, uint32_t K_want, MultiPtr *out)
{
    MultiPtr ms(new (std::nothrow) gm::MultiState());
    GM_CHECK(ms, GM_ERR_NOMEM, "gm_page_rank_multi: out of host memory");
    const uint32_t P = (uint32_t)devs.size();
    ms->devs = devs, ms->damping = damping_factor, ms->engine = engine_env, ms->P = P, ms->n = n, ms->distinct = distinct;
    ms->env_hash = multi_env_hash();
    ms->bounds = bounds_in;
    gm::PhaseTimer timer((hipStream_t)0);
    const std::vector<uint32_t> &bounds = ms->bounds;
    // a sweep in pieces needs propagation-blocking engines on every rank: every slice large enough for AUTO to pick
    // one (pagerank.hip: 2^24 edges), or forced by GM_MULTI_ENGINE=pb
    uint64_t min_edges = ~0ull;
    for (uint32_t p = 0; p < P; ++p)
        min_edges = std::min<uint64_t>(min_edges, in[p].edges);
    ms->pieces = K_want > 1 && (engine_env == GM_PR_ENGINE_PB || (engine_env == GM_PR_ENGINE_AUTO && min_edges >= (1ull << 24)));
    const uint32_t K = ms->pieces ? K_want : 1;
    ms->K = K;
    const int engine = ms->pieces ? GM_PR_ENGINE_PB : engine_env;
    // row groups of every rank (global ids), cut at multiples of MG_ROW_ALIGN local rows
    std::vector<uint32_t> splits((size_t)P * (K + 1));
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t rows = bounds[p + 1] - bounds[p];
        for (uint32_t k = 0; k <= K; ++k) {
            uint64_t cut = k == K ? rows : ((uint64_t)rows * k / K + MG_ROW_ALIGN - 1) / MG_ROW_ALIGN * MG_ROW_ALIGN;
            splits[(size_t)p * (K + 1) + k] = bounds[p] + (uint32_t)std::min<uint64_t>(cut, rows);
        }
    }

    // ---- per rank, on its own device: who is gathered (flag / pos), the exchange layout, the node map -------------------
    struct Local {
        DevBuf flag, pos, node_map;
    };
    std::vector<Local> loc(P);
    std::vector<uint32_t> pos_at(splits.size());
    ms->strides.assign(K, 0), ms->region_off.assign(K, 0);
    for (uint32_t p = 0; p < P; ++p) {
        DeviceGuard g(devs[p]);
        GM_TRY(loc[p].flag.alloc(((size_t)n + 1) * 4));
        GM_TRY(loc[p].pos.alloc(((size_t)n + 1) * 4));
        hipLaunchKernelGGL(mg_has_out_degree_kernel, dim3(mg_grid((uint64_t)n + 1)), dim3(256), 0, 0, in[p].outdeg_full.as<uint32_t>(),
                           n, loc[p].flag.as<uint32_t>());
        GM_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, loc[p].flag.as<uint32_t>(), loc[p].pos.as<uint32_t>(), 0u, (size_t)n + 1,
                                       rocprim::plus<uint32_t>(), (hipStream_t)0));
        DevBuf tmp;
        GM_TRY(tmp.alloc(tmp_bytes));
        GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, loc[p].flag.as<uint32_t>(), loc[p].pos.as<uint32_t>(), 0u, (size_t)n + 1,
                                       rocprim::plus<uint32_t>(), (hipStream_t)0));
        GM_HIP(hipDeviceSynchronize());
        if (p == 0) { // the layout's numbers: the same on every rank, read once
            for (size_t i = 0; i < splits.size(); ++i)
                GM_HIP(hipMemcpy(&pos_at[i], loc[0].pos.as<uint32_t>() + splits[i], 4, hipMemcpyDeviceToHost));
            uint64_t off = 0;
            for (uint32_t k = 0; k < K; ++k) {
                uint32_t most = 1;
                for (uint32_t q = 0; q < P; ++q)
                    most = std::max(most, pos_at[(size_t)q * (K + 1) + k + 1] - pos_at[(size_t)q * (K + 1) + k]);
                // whole source tiles per rank when the vector is consumed region by region; float4-aligned otherwise
                const uint32_t unit = K > 1 ? MG_SOURCE_TILE : 4u;
                ms->strides[k] = (most + unit - 1) / unit * unit;
                ms->region_off[k] = (uint32_t)off;
                off += ms->strides[k];
            }
            GM_CHECK(off * P < (1ull << 32), GM_ERR_RANGE, "gm_page_rank_multi: exchange vector exceeds u32");
            ms->rank_stride = (uint32_t)off;
            ms->x_len = off * P;
        }
        DevBuf d_splits, d_region;
        GM_TRY(loc[p].node_map.alloc((size_t)n * 4));
        GM_TRY(d_splits.alloc(splits.size() * 4));
        GM_TRY(d_region.alloc((size_t)K * 4));
        GM_HIP(hipMemcpy(d_splits.p, splits.data(), splits.size() * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemcpy(d_region.p, ms->region_off.data(), (size_t)K * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mg_node_map_parts_kernel, dim3(mg_grid(n)), dim3(256), 0, 0, loc[p].flag.as<uint32_t>(),
                           loc[p].pos.as<uint32_t>(), d_splits.as<uint32_t>(), P, K, ms->rank_stride, d_region.as<uint32_t>(), n,
                           loc[p].node_map.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
    }
    const uint64_t x_len = ms->x_len;
    timer.done("multi: partition + exchange layout (%u ranks, %u regions)", P, K);
    // Which nodes have no in-edges at all?  Every rank knows it of its own rows; the plan builder's rule for rows of constant terms
    // (GM_PB_HUB_LEAVES) needs it of every SOURCE, so that a slice flags the rows the whole graph's plan flags and the partitioned
    // run's bits stay the single engine's.  n bytes through the host (any topology), only when the rule is on.
    std::vector<uint8_t> has_in;
    if (gm::hub_leaves_threshold()) {
        has_in.assign((size_t)n, 0);
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t rows = bounds[p + 1] - bounds[p];
            if (!rows)
                continue;
            DeviceGuard g(devs[p]);
            DevBuf d;
            GM_TRY(d.alloc(rows));
            hipLaunchKernelGGL(mg_has_in_kernel, dim3(mg_grid(rows)), dim3(256), 0, 0, in[p].off.as<uint32_t>(), rows, d.as<uint8_t>());
            GM_HIP(hipGetLastError());
            GM_HIP(hipMemcpy(has_in.data() + bounds[p], d.p, rows, hipMemcpyDeviceToHost));
        }
    }

    // ---- one rank per device: row slice, exchange buffers, engine ------------------------------------------
    for (uint32_t p = 0; p < P; ++p) {
        ms->ranks.emplace_back(new Rank());
        Rank &r = *ms->ranks.back();
        r.device = devs[p];
        r.lo = bounds[p], r.hi = bounds[p + 1];
        const uint32_t rows = r.hi - r.lo;
        const uint64_t cnt = in[p].edges;
        DeviceGuard g(r.device);
        GM_HIP(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
        GM_HIP(hipStreamCreateWithFlags(&r.cst, hipStreamNonBlocking));
        for (uint32_t k = 0; k < K; ++k) {
            GM_HIP(hipEventCreateWithFlags(&r.ev_ready[k], hipEventDisableTiming));
            GM_HIP(hipEventCreateWithFlags(&r.ev_done[0][k], hipEventDisableTiming));
            GM_HIP(hipEventCreateWithFlags(&r.ev_done[1][k], hipEventDisableTiming));
        }
        GM_TRY(r.outdeg.alloc((size_t)(rows ? rows : 1) * 4));
        if (rows)
            GM_HIP(hipMemcpy(r.outdeg.p, in[p].outdeg_full.as<uint32_t>() + r.lo, (size_t)rows * 4, hipMemcpyDeviceToDevice));
        for (uint32_t k = 0; k < K; ++k) { // rows[rank among the group's senders] = v - (rank's first row)
            const uint32_t g_lo = splits[(size_t)p * (K + 1) + k], g_hi = splits[(size_t)p * (K + 1) + k + 1];
            r.send_count[k] = pos_at[(size_t)p * (K + 1) + k + 1] - pos_at[(size_t)p * (K + 1) + k];
            GM_TRY(r.send_rows[k].alloc((size_t)r.send_count[k] * 4));
            if (g_hi > g_lo)
                hipLaunchKernelGGL(mg_send_rows_kernel, dim3(mg_grid(g_hi - g_lo)), dim3(256), 0, 0, loc[p].flag.as<uint32_t>(),
                                   loc[p].pos.as<uint32_t>(), g_lo, g_hi, r.lo, r.send_rows[k].as<uint32_t>());
            GM_HIP(hipGetLastError());
        }
        if (cnt) // targets into the exchange index space
            hipLaunchKernelGGL(mg_map_targets_kernel, dim3(mg_grid(cnt)), dim3(256), 0, 0, in[p].tgt.as<uint32_t>(), cnt,
                               loc[p].node_map.as<uint32_t>());
        GM_HIP(hipGetLastError());
        DevBuf slot_flags;
        if (!has_in.empty() && n) {
            DevBuf d_has_in;
            GM_TRY(d_has_in.alloc((size_t)n));
            GM_TRY(slot_flags.alloc((size_t)(x_len ? x_len : 1)));
            GM_HIP(hipMemcpy(d_has_in.p, has_in.data(), (size_t)n, hipMemcpyHostToDevice));
            GM_HIP(hipMemset(slot_flags.p, 0, (size_t)(x_len ? x_len : 1)));
            hipLaunchKernelGGL(mg_slot_flags_kernel, dim3(mg_grid(n)), dim3(256), 0, 0, loc[p].node_map.as<uint32_t>(), d_has_in.as<uint8_t>(),
                               n, slot_flags.as<uint8_t>());
            GM_HIP(hipGetLastError());
            GM_HIP(hipDeviceSynchronize());
        }
        GM_HIP(hipDeviceSynchronize());
        loc[p].flag.release(), loc[p].pos.release(), loc[p].node_map.release();
        in[p].outdeg_full.release();
        // hand the arrays to an owning handle: the Rank keeps the buffers alive through it
        gm_csr *c = new (std::nothrow) gm_csr();
        GM_CHECK(c, GM_ERR_NOMEM, "gm_page_rank_multi: out of host memory");
        c->n = rows, c->m = cnt, c->device = r.device, c->owns = true;
        c->own_offsets = std::move(in[p].off);
        c->own_targets = std::move(in[p].tgt);
        c->offsets = c->own_offsets.as<uint32_t>();
        c->targets = c->own_targets.as<uint32_t>();
        r.rows = c;
        if (slot_flags.p && x_len)
            GM_TRY(gm_csr_set_source_flags(c, (uint64_t)slot_flags.p, x_len));
        slot_flags.release();
        GM_TRY(r.scores.alloc((size_t)(rows ? rows : 1) * 4));
        GM_TRY(r.x_loc.alloc((size_t)(rows ? rows : 1) * 4));
        GM_TRY(r.x[0].alloc((size_t)x_len * 4));
        GM_TRY(r.x[1].alloc((size_t)x_len * 4));
        GM_TRY(r.err.alloc(8));
        GM_HIP(hipMemset(r.x[0].p, 0, (size_t)x_len * 4));
        GM_HIP(hipMemset(r.x[1].p, 0, (size_t)x_len * 4));
        for (uint32_t k = 0; k < K; ++k) {
            GM_TRY(r.x_send[k].alloc((size_t)ms->strides[k] * 4));
            GM_HIP(hipMemset(r.x_send[k].p, 0, (size_t)ms->strides[k] * 4));
        }
        GM_HIP(hipMemset(r.err.p, 0, 8));
        GM_HIP(hipDeviceSynchronize());
        GM_TRY(gm_pr_create_with(r.rows, n, r.lo, x_len, (uint64_t)r.outdeg.p, damping_factor, engine, &r.pr));
        if (ms->pieces) {
            GM_CHECK(gm_pr_engine(r.pr) == GM_PR_ENGINE_PB, GM_ERR_INVALID, "gm_page_rank_multi: rank %u did not get a propagation-blocking engine", p);
            uint64_t sp[MG_MAX_PARTS + 1];
            for (uint32_t k = 0; k <= K; ++k)
                sp[k] = splits[(size_t)p * (K + 1) + k] - r.lo;
            GM_TRY(gm_pr_set_parts(r.pr, sp, K));
            // region k of the vector = row group k of every rank: P ranges, propagated in one launch
            std::vector<uint64_t> x_lo, x_hi;
            std::vector<uint32_t> reg;
            for (uint32_t q = 0; q < P; ++q)
                for (uint32_t k = 0; k < K; ++k) {
                    x_lo.push_back((uint64_t)q * ms->rank_stride + ms->region_off[k]);
                    x_hi.push_back(x_lo.back() + ms->strides[k]);
                    reg.push_back(k);
                }
            GM_TRY(gm_pr_set_bin_regions(r.pr, x_lo.data(), x_hi.data(), reg.data(), x_lo.size(), K));
        }
    }
    timer.done("multi: row slices + engines");
    if (distinct) {
        GM_TRY(rccl_get(&ms->comms.rc));
        ms->comms.c.assign(P, nullptr);
        GM_NCCL(ms->comms.rc, ms->comms.rc->CommInitAll(ms->comms.c.data(), (int)P, devs.data()));
        timer.done("multi: ncclCommInitAll");
    }
    GM_TRY(ms->herr.alloc((size_t)P * 8));
    *out = std::move(ms);
    return GM_OK;
}

// gm_page_rank_multi: both CSRs whole on one device: the reference's partitioner on the host (from the in-offsets), then
// every rank's input is cut there and moved to the rank's device
int multi_build(const gm_csr *out_csr, const gm_csr *in_csr, const std::vector<int> &devs, bool distinct, float damping_factor,
                int engine_env, uint32_t K_want, MultiPtr *out)
{
    const uint32_t n = (uint32_t)in_csr->n, P = (uint32_t)devs.size();
    const int src_dev = in_csr->device;
    std::vector<uint32_t> off_host((size_t)n + 1);
    DevBuf od_full;
    {
        DeviceGuard g(src_dev);
        GM_HIP(hipMemcpy(off_host.data(), in_csr->offsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
        GM_TRY(od_full.alloc((size_t)n * 4));
        hipLaunchKernelGGL(mg_out_degree_kernel, dim3(mg_grid(n)), dim3(256), 0, 0, out_csr->offsets, 0u, n, od_full.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
    }
    const std::vector<uint32_t> bounds = greedy_in_degree_bounds(off_host, P);
    std::vector<RankInput> in(P);
    for (uint32_t p = 0; p < P; ++p) {
        RankInput &ri = in[p];
        ri.device = devs[p];
        const uint32_t lo = bounds[p], rows = bounds[p + 1] - bounds[p];
        const uint64_t e0 = off_host[lo], cnt = (uint64_t)off_host[lo + rows] - e0;
        ri.edges = cnt;
        DeviceGuard g(ri.device);
        GM_TRY(ri.off.alloc(((size_t)rows + 1) * 4));
        GM_TRY(ri.tgt.alloc((size_t)cnt * 4));
        GM_TRY(ri.outdeg_full.alloc((size_t)n * 4));
        GM_HIP(mg_copy(ri.off.p, ri.device, in_csr->offsets + lo, src_dev, ((size_t)rows + 1) * 4));
        GM_HIP(mg_copy(ri.tgt.p, ri.device, in_csr->targets + e0, src_dev, (size_t)cnt * 4));
        GM_HIP(mg_copy(ri.outdeg_full.p, ri.device, od_full.p, src_dev, (size_t)n * 4));
        hipLaunchKernelGGL(mg_rebase_kernel, dim3(mg_grid((uint64_t)rows + 1)), dim3(256), 0, 0, ri.off.as<uint32_t>(), rows + 1,
                           (uint32_t)e0);
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
    }
    GM_TRY(multi_build_from(in, bounds, n, devs, distinct, damping_factor, engine_env, K_want, out));
    (*out)->out_csr = out_csr;
    return GM_OK;
}

// compacts the out_scores of row group k on every rank (its sweep stream) and starts their all-gather into region k
// of x[buf] on the exchange streams; nothing waits on the host
int multi_start_exchange(gm::MultiState &ms, int buf, uint32_t k)
{
    const uint32_t P = ms.P;
    for (auto &rp : ms.ranks) {
        Rank &r = *rp;
        DeviceGuard g(r.device);
        if (!ms.distinct) // copies stand in for the collective: x_send[k] is still being read until EVERY receiver of its last use is done
            for (auto &other : ms.ranks)
                GM_HIP(hipStreamWaitEvent(r.st, other->ev_done[1 - buf][k], 0));
        if (r.send_count[k])
            hipLaunchKernelGGL(mg_compact_kernel, dim3(mg_grid(r.send_count[k])), dim3(256), 0, r.st, r.x_loc.as<float>(),
                               r.send_rows[k].as<uint32_t>(), r.send_count[k], r.x_send[k].as<float>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipEventRecord(r.ev_ready[k], r.st));
    }
    const size_t stride = ms.strides[k];
    if (ms.distinct) {
        const Rccl *rc = ms.comms.rc;
        for (auto &rp : ms.ranks) {
            DeviceGuard g(rp->device);
            GM_HIP(hipStreamWaitEvent(rp->cst, rp->ev_ready[k], 0));
        }
        GM_NCCL(rc, rc->GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (uint32_t q = 0; q < P && bad == ncclSuccess; ++q) {
            Rank &r = *ms.ranks[q];
            if (ms.K == 1) { // one region: the ranks' stretches are exactly what an all-gather lays out
                bad = rc->AllGather(r.x_send[k].p, r.x[buf].p, stride, ncclFloat32, ms.comms.c[q], r.cst);
                continue;
            }
            // row group k of rank p goes to p * rank_stride + region_off[k] on every rank: P broadcasts in one group
            for (uint32_t p = 0; p < P && bad == ncclSuccess; ++p)
                bad = rc->Broadcast(r.x_send[k].p, r.x[buf].as<float>() + (size_t)p * ms.rank_stride + ms.region_off[k], stride,
                                    ncclFloat32, (int)p, ms.comms.c[q], r.cst);
        }
        const ncclResult_t end = rc->GroupEnd(); // always closed, also after a failed call inside the group
        if (bad != ncclSuccess || end != ncclSuccess) {
            gm::set_error("gm_page_rank_multi: ncclAllGather failed: %s", rc->GetErrorString(bad != ncclSuccess ? bad : end));
            return GM_ERR_HIP;
        }
    } else { // virtual ranks on shared devices: the same data movement with copies, on the receivers' exchange streams
        for (uint32_t q = 0; q < P; ++q) {
            Rank &dst = *ms.ranks[q];
            DeviceGuard g(dst.device);
            // like a collective, nothing moves before every rank has arrived — the receiver included: its own sweep has
            // then left the buffer that is overwritten here
            for (uint32_t p = 0; p < P; ++p)
                GM_HIP(hipStreamWaitEvent(dst.cst, ms.ranks[p]->ev_ready[k], 0));
            for (uint32_t p = 0; p < P; ++p) {
                Rank &src = *ms.ranks[p];
                float *to = dst.x[buf].as<float>() + (size_t)p * ms.rank_stride + ms.region_off[k];
                if (dst.device == src.device)
                    GM_HIP(hipMemcpyAsync(to, src.x_send[k].p, stride * 4, hipMemcpyDeviceToDevice, dst.cst));
                else
                    GM_HIP(hipMemcpyPeerAsync(to, dst.device, src.x_send[k].p, src.device, stride * 4, dst.cst));
            }
        }
    }
    for (auto &rp : ms.ranks) {
        DeviceGuard g(rp->device);
        GM_HIP(hipEventRecord(rp->ev_done[buf][k], rp->cst));
    }
    return GM_OK;
}

int multi_run(gm::MultiState &ms, uint64_t max_iterations, double tolerance, float *scores_out, uint64_t *iterations_out,
              double *error_out)
{
    const uint32_t P = ms.P, K = ms.K;
    uint64_t host_syncs = 0;
    for (auto &rp : ms.ranks) {
        DeviceGuard g(rp->device);
        GM_TRY(gm_pr_init(rp->pr, (uint64_t)rp->scores.p, (uint64_t)rp->x_loc.p, rp->st));
    }
    int cur = 0;
    for (uint32_t k = 0; k < K; ++k)
        GM_TRY(multi_start_exchange(ms, cur, k));
    uint64_t iter = 0;
    double err = 0.0;
    const bool can_stop_early = tolerance > 0.0;
    for (;;) {
        // a region of x[cur] is propagated as soon as it has landed; then the row groups, each followed by the start
        // of its region's exchange into x[1 - cur] — which travels under the work on the other groups
        if (ms.pieces) {
            for (uint32_t k = 0; k < K; ++k)
                for (auto &rp : ms.ranks) {
                    DeviceGuard g(rp->device);
                    GM_HIP(hipStreamWaitEvent(rp->st, rp->ev_done[cur][k], 0));
                    GM_TRY(gm_pr_sweep_bin_region(rp->pr, (uint64_t)rp->x[cur].p, k, rp->st));
                }
            for (uint32_t k = 0; k < K; ++k) {
                for (auto &rp : ms.ranks) {
                    DeviceGuard g(rp->device);
                    GM_TRY(gm_pr_sweep_accum(rp->pr, (uint64_t)rp->x[cur].p, (uint64_t)rp->x_loc.p, (uint64_t)rp->scores.p, k,
                                             k == 0 ? 1 : 0, rp->st));
                }
                if (iter + 1 != max_iterations || can_stop_early) // the last sweep's out_scores are not needed by anyone
                    GM_TRY(multi_start_exchange(ms, 1 - cur, k));
            }
            for (auto &rp : ms.ranks) {
                DeviceGuard g(rp->device);
                GM_TRY(gm_pr_sweep_fixup(rp->pr, (uint64_t)rp->x_loc.p, (uint64_t)rp->scores.p, (uint64_t)rp->err.p, rp->st));
            }
        } else {
            for (auto &rp : ms.ranks) {
                DeviceGuard g(rp->device);
                GM_HIP(hipStreamWaitEvent(rp->st, rp->ev_done[cur][0], 0));
                GM_TRY(gm_pr_sweep(rp->pr, (uint64_t)rp->x[cur].p, (uint64_t)rp->x_loc.p, (uint64_t)rp->scores.p,
                                   (uint64_t)rp->err.p, rp->st));
            }
            if (iter + 1 != max_iterations || can_stop_early)
                GM_TRY(multi_start_exchange(ms, 1 - cur, 0));
        }
        iter += 1;
        const bool last = iter == max_iterations;
        if (can_stop_early || last) {
            for (uint32_t p = 0; p < P; ++p) {
                DeviceGuard g(ms.ranks[p]->device);
                GM_HIP(hipMemcpyAsync(ms.herr.as<double>() + p, ms.ranks[p]->err.p, 8, hipMemcpyDeviceToHost, ms.ranks[p]->st));
            }
            for (auto &rp : ms.ranks) {
                DeviceGuard g(rp->device);
                GM_HIP(hipStreamSynchronize(rp->st));
            }
            ++host_syncs;
            err = 0.0;
            for (uint32_t p = 0; p < P; ++p) // rank order: the same bits on every run
                err += ms.herr.as<double>()[p];
            if (err < tolerance || last)
                break;
        }
        cur = 1 - cur;
    }
    for (auto &rp : ms.ranks) {
        DeviceGuard g(rp->device);
        if (rp->hi > rp->lo)
            GM_HIP(hipMemcpyAsync(scores_out + rp->lo, rp->scores.p, (size_t)(rp->hi - rp->lo) * 4, hipMemcpyDeviceToHost, rp->st));
    }
    for (auto &rp : ms.ranks) { // exchanges still in flight (started for a sweep that the stop rule cancelled) included
        DeviceGuard g(rp->device);
        GM_HIP(hipStreamSynchronize(rp->st));
        GM_HIP(hipStreamSynchronize(rp->cst));
    }
    if (gm::log_enabled())
        fprintf(stderr, "[graph_mi355x] multi: %llu sweeps on %u ranks, %u region(s)%s, %llu host synchronisation(s) before the result copy\n",
                (unsigned long long)iter, P, K, ms.pieces ? " overlapped with the work" : "", (unsigned long long)host_syncs);
    *iterations_out = iter;
    *error_out = err;
    return GM_OK;
}

} // namespace

GM_API int gm_page_rank_multi(const gm_csr *out_csr, const gm_csr *in_csr, const int *devices, uint32_t n_devices,
                              uint64_t max_iterations, double tolerance, float damping_factor, float *scores_out,
                              uint64_t *iterations_out, double *error_out)
{
    GM_CHECK(out_csr && in_csr && iterations_out && error_out, GM_ERR_INVALID, "gm_page_rank_multi: null argument");
    GM_CHECK(out_csr->n == in_csr->n && out_csr->m == in_csr->m && out_csr->device == in_csr->device, GM_ERR_INVALID,
             "gm_page_rank_multi: the two CSRs are not the out- and in-lists of one graph on one device");
    GM_CHECK(n_devices >= 1 && n_devices <= 64, GM_ERR_INVALID, "gm_page_rank_multi: n_devices %u not in [1, 64]", n_devices);
    GM_CHECK(max_iterations != 0 || tolerance > 0.0, GM_ERR_INVALID,
             "gm_page_rank_multi: max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)");
    const uint32_t n = (uint32_t)in_csr->n, P = n_devices;
    if (n == 0) {
        *iterations_out = 1;
        *error_out = 0.0;
        return GM_OK;
    }
    GM_CHECK(scores_out, GM_ERR_INVALID, "gm_page_rank_multi: scores_out is null");
    int visible = 0;
    GM_HIP(hipGetDeviceCount(&visible));
    std::vector<int> devs(P);
    bool distinct = true;
    for (uint32_t p = 0; p < P; ++p) {
        devs[p] = devices ? devices[p] : (int)p;
        GM_CHECK(devs[p] >= 0 && devs[p] < visible, GM_ERR_INVALID, "gm_page_rank_multi: device %d of %d visible", devs[p],
                 visible);
        for (uint32_t q = 0; q < p; ++q)
            distinct = distinct && devs[q] != devs[p];
    }
    // knobs, read once per call (never inside the sweep loop)
    const int engine = [] {
        const char *v = getenv("GM_MULTI_ENGINE"); // tests: "pb" / "pull"; default: by the size of each slice
        return v && v[0] == 'p' && v[1] == 'b' ? GM_PR_ENGINE_PB : v && v[0] == 'p' ? GM_PR_ENGINE_PULL : GM_PR_ENGINE_AUTO;
    }();
    uint32_t K = 2; // regions of the exchange that overlap with the work (GM_MULTI_PARTS; 1 = one all-gather per sweep)
    if (const char *v = getenv("GM_MULTI_PARTS"))
        K = (uint32_t)atoi(v);
    K = K < 1 ? 1 : (K > MG_MAX_PARTS ? MG_MAX_PARTS : K);

    // the resident state of the previous call on this graph, if it was made for the same run
    MultiPtr ms;
    {
        std::lock_guard<std::mutex> lock(in_csr->cache_mu);
        if (in_csr->multi && in_csr->multi->out_csr == out_csr && in_csr->multi->devs == devs &&
            in_csr->multi->damping == damping_factor && in_csr->multi->engine == engine && in_csr->multi->env_hash == multi_env_hash() &&
            (in_csr->multi->K == K || (!in_csr->multi->pieces && in_csr->multi->K == 1)) && !getenv("GM_MULTI_NOCACHE") &&
            !(getenv("GM_PB_NOCACHE") && atoi(getenv("GM_PB_NOCACHE")))) // measurement runs that switch plan knobs build afresh
            ms = std::move(in_csr->multi);
    }
    if (!ms)
        GM_TRY(multi_build(out_csr, in_csr, devs, distinct, damping_factor, engine, K, &ms));
    const int rc = multi_run(*ms, max_iterations, tolerance, scores_out, iterations_out, error_out);
    if (rc == GM_OK && !(getenv("GM_PB_NOCACHE") && atoi(getenv("GM_PB_NOCACHE")))) { // parked for the next call (a failed run is torn down)
        MultiPtr old; // the state of another run (other devices, another K) makes room: destroyed outside the lock
        {
            std::lock_guard<std::mutex> lock(in_csr->cache_mu);
            old = std::move(in_csr->multi);
            in_csr->multi = std::move(ms);
        }
    }
    return rc;
}

// The same run from per-device pieces: no device ever holds the whole graph.  in_slices[p]: the rows [bounds[p],
// bounds[p + 1]) of the in-CSR on devices[p] (n_local rows, offsets from 0, targets as GLOBAL node ids, Sorted / Deduplicated
// layout for the reference's summation order); d_out_degree_full[p]: u32[n] on devices[p], the out-degree of every node
// (a host that builds the pieces separately gets it by an all-reduce of per-piece histograms).  The inputs are copied, not
// consumed.  The partition is the caller's: in_degree_partition's greedy ranges (graph_ops.rs:431-439,479-509) give the
// same bits as gm_page_rank_multi on the whole graph.
GM_API int gm_page_rank_multi_slices(const gm_csr *const *in_slices, const uint64_t *bounds, const uint64_t *d_out_degree_full,
                                     uint64_t n, const int *devices, uint32_t n_devices, uint64_t max_iterations, double tolerance,
                                     float damping_factor, float *scores_out, uint64_t *iterations_out, double *error_out)
{
    GM_CHECK(in_slices && bounds && d_out_degree_full && iterations_out && error_out, GM_ERR_INVALID,
             "gm_page_rank_multi_slices: null argument");
    GM_CHECK(n_devices >= 1 && n_devices <= 64, GM_ERR_INVALID, "gm_page_rank_multi_slices: n_devices %u not in [1, 64]", n_devices);
    GM_CHECK(n < (1ull << 32), GM_ERR_RANGE, "gm_page_rank_multi_slices: %llu nodes", (unsigned long long)n);
    GM_CHECK(max_iterations != 0 || tolerance > 0.0, GM_ERR_INVALID,
             "gm_page_rank_multi_slices: max_iterations == 0 with tolerance <= 0 never terminates (reference: infinite loop)");
    if (n == 0) {
        *iterations_out = 1;
        *error_out = 0.0;
        return GM_OK;
    }
    GM_CHECK(scores_out, GM_ERR_INVALID, "gm_page_rank_multi_slices: scores_out is null");
    const uint32_t P = n_devices;
    int visible = 0;
    GM_HIP(hipGetDeviceCount(&visible));
    std::vector<int> devs(P);
    std::vector<uint32_t> b32(P + 1);
    bool distinct = true;
    GM_CHECK(bounds[0] == 0 && bounds[P] == n, GM_ERR_INVALID, "gm_page_rank_multi_slices: bounds must run from 0 to n");
    for (uint32_t p = 0; p <= P; ++p)
        b32[p] = (uint32_t)bounds[p];
    for (uint32_t p = 0; p < P; ++p) {
        devs[p] = devices ? devices[p] : (int)p;
        GM_CHECK(devs[p] >= 0 && devs[p] < visible, GM_ERR_INVALID, "gm_page_rank_multi_slices: device %d of %d visible", devs[p],
                 visible);
        for (uint32_t q = 0; q < p; ++q)
            distinct = distinct && devs[q] != devs[p];
        GM_CHECK(bounds[p] <= bounds[p + 1], GM_ERR_INVALID, "gm_page_rank_multi_slices: bounds must ascend");
        GM_CHECK(in_slices[p] && in_slices[p]->n == bounds[p + 1] - bounds[p] && in_slices[p]->device == devs[p] && d_out_degree_full[p],
                 GM_ERR_INVALID, "gm_page_rank_multi_slices: piece %u is not the rows [%llu, %llu) on device %d", p,
                 (unsigned long long)bounds[p], (unsigned long long)bounds[p + 1], devs[p]);
    }
    const int engine = [] {
        const char *v = getenv("GM_MULTI_ENGINE");
        return v && v[0] == 'p' && v[1] == 'b' ? GM_PR_ENGINE_PB : v && v[0] == 'p' ? GM_PR_ENGINE_PULL : GM_PR_ENGINE_AUTO;
    }();
    uint32_t K = 2;
    if (const char *v = getenv("GM_MULTI_PARTS"))
        K = (uint32_t)atoi(v);
    K = K < 1 ? 1 : (K > MG_MAX_PARTS ? MG_MAX_PARTS : K);
    std::vector<RankInput> in(P);
    for (uint32_t p = 0; p < P; ++p) {
        RankInput &ri = in[p];
        const gm_csr *sl = in_slices[p];
        ri.device = devs[p];
        ri.edges = sl->m;
        DeviceGuard g(ri.device);
        // a piece is the caller's: its offsets must run from 0 to its edge count and its targets must be global ids below n
        // (they index the n-sized node map on this device) — GM_ERR_RANGE instead of a memory fault
        {
            uint32_t ends[2] = {0u, 0u};
            GM_HIP(hipMemcpy(&ends[0], sl->offsets, 4, hipMemcpyDeviceToHost));
            GM_HIP(hipMemcpy(&ends[1], sl->offsets + sl->n, 4, hipMemcpyDeviceToHost));
            GM_CHECK(ends[0] == 0u && ends[1] == sl->m, GM_ERR_INVALID,
                     "gm_page_rank_multi_slices: piece %u: offsets run from %u to %u, the piece has %llu edges", p, ends[0], ends[1],
                     (unsigned long long)sl->m);
            if (sl->m) {
                DevBuf d_max;
                GM_TRY(d_max.alloc(4));
                GM_HIP(hipMemset(d_max.p, 0, 4));
                unsigned mg = gm::div_up(sl->m, 256);
                hipLaunchKernelGGL(mg_max_target_kernel, dim3(mg > 4096 ? 4096 : mg), dim3(256), 0, 0, sl->targets, sl->m, d_max.as<uint32_t>());
                GM_HIP(hipGetLastError());
                uint32_t mx = 0;
                GM_HIP(hipMemcpy(&mx, d_max.p, 4, hipMemcpyDeviceToHost));
                GM_CHECK(mx < n, GM_ERR_RANGE, "gm_page_rank_multi_slices: piece %u names node %u of %llu (targets are GLOBAL ids)", p, mx,
                         (unsigned long long)n);
            }
        }
        GM_TRY(ri.off.alloc(((size_t)sl->n + 1) * 4));
        GM_TRY(ri.tgt.alloc((size_t)sl->m * 4));
        GM_TRY(ri.outdeg_full.alloc((size_t)n * 4));
        GM_HIP(hipMemcpy(ri.off.p, sl->offsets, ((size_t)sl->n + 1) * 4, hipMemcpyDeviceToDevice));
        if (sl->m)
            GM_HIP(hipMemcpy(ri.tgt.p, sl->targets, (size_t)sl->m * 4, hipMemcpyDeviceToDevice));
        GM_HIP(hipMemcpy(ri.outdeg_full.p, reinterpret_cast<const void *>(d_out_degree_full[p]), (size_t)n * 4, hipMemcpyDeviceToDevice));
    }
    MultiPtr ms;
    GM_TRY(multi_build_from(in, b32, (uint32_t)n, devs, distinct, damping_factor, engine, K, &ms));
    return multi_run(*ms, max_iterations, tolerance, scores_out, iterations_out, error_out);
}

namespace gm {
void warm_multi() // (common.hpp: the code object of this file, loaded ahead of an algorithm's first call)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&mg_has_out_degree_kernel)) != hipSuccess)
        (void)hipGetLastError();
}
} // namespace gm
