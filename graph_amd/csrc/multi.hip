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
//               page_rank.rs:78,158): they are numbered rank-major with a fixed per-rank stride, every rank's
//               targets are rewritten into that index space once, and the all-gather output is consumed directly
//               as the next sweep's x vector (an all-gather moves half the bytes of the all-reduce of a
//               zero-padded vector and gives the same result)
//   per sweep   local sweep kernels (the same engines as the single-GPU path) -> compaction gather -> ncclAllGather
//               on the rank's stream -> f64 error partials summed on the host in rank order (deterministic)
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

namespace {

using namespace gm;

// ---- RCCL, bound at run time ---------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
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
        t.GroupStart = reinterpret_cast<decltype(t.GroupStart)>(dlsym(h, "ncclGroupStart"));
        t.GroupEnd = reinterpret_cast<decltype(t.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        GM_CHECK(t.CommInitAll && t.CommDestroy && t.AllGather && t.GroupStart && t.GroupEnd && t.GetErrorString,
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
__global__ void mg_has_out_kernel(const uint32_t *__restrict__ out_off, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u <= n; u += stride)
        flag[u] = (u < n && out_off[u + 1] > out_off[u]) ? 1u : 0u;
}

// node v of rank p (bounds[p] <= v < bounds[p+1]) with out-edges -> slot p * stride + (its rank among the
// rank's nodes with out-edges); nodes without out-edges are never a target of an in-list
__global__ void mg_node_map_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                   const uint32_t *__restrict__ bounds, uint32_t parts, uint32_t stride, uint32_t n,
                                   uint32_t *__restrict__ node_map)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += s) {
        uint32_t p = 0; // parts <= 64: a short scan
        while (p + 1 < parts && v >= bounds[p + 1])
            ++p;
        node_map[v] = flag[v] ? p * stride + (pos[v] - pos[bounds[p]]) : 0xFFFFFFFFu;
    }
}

// local rows (of [lo, hi)) that have out-edges, in order: what the rank contributes to the exchange
__global__ void mg_send_rows_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, uint32_t lo,
                                    uint32_t hi, uint32_t *__restrict__ rows)
{
    const uint32_t s = gridDim.x * blockDim.x;
    for (uint32_t v = lo + blockIdx.x * blockDim.x + threadIdx.x; v < hi; v += s)
        if (flag[v])
            rows[pos[v] - pos[lo]] = v - lo;
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

struct Rank {
    int device = 0;
    uint32_t lo = 0, hi = 0, send_count = 0;
    hipStream_t st = nullptr;
    gm_csr *rows = nullptr; // row slice with targets in exchange index space (lives on `device`)
    gm_pr *pr = nullptr;
    DevBuf outdeg, scores, x_loc, x[2], x_send, send_rows, err;
    ~Rank()
    {
        DeviceGuard g(device);
        if (pr)
            gm_pr_destroy(pr);
        if (rows)
            gm_csr_free(rows);
        if (st)
            (void)hipStreamDestroy(st);
        outdeg.release(), scores.release(), x_loc.release(), x[0].release(), x[1].release(), x_send.release(),
            send_rows.release(), err.release();
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
    const int src_dev = in_csr->device;

    // ---- partition + exchange layout (on the device that holds the graph) --------------------------------
    std::vector<uint32_t> off_host((size_t)n + 1);
    std::vector<uint32_t> bounds, pos_at(P + 1);
    DevBuf flag, pos, d_bounds, node_map;
    uint32_t stride = 1;
    {
        DeviceGuard g(src_dev);
        GM_HIP(hipMemcpy(off_host.data(), in_csr->offsets, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
        bounds = greedy_in_degree_bounds(off_host, P);
        GM_TRY(flag.alloc(((size_t)n + 1) * 4));
        GM_TRY(pos.alloc(((size_t)n + 1) * 4));
        GM_TRY(d_bounds.alloc(((size_t)P + 1) * 4));
        GM_TRY(node_map.alloc((size_t)n * 4));
        hipLaunchKernelGGL(mg_has_out_kernel, dim3(mg_grid((uint64_t)n + 1)), dim3(256), 0, 0, out_csr->offsets, n,
                           flag.as<uint32_t>());
        GM_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n + 1,
                                       rocprim::plus<uint32_t>(), (hipStream_t)0));
        DevBuf tmp;
        GM_TRY(tmp.alloc(tmp_bytes));
        GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)n + 1,
                                       rocprim::plus<uint32_t>(), (hipStream_t)0));
        for (uint32_t p = 0; p <= P; ++p)
            GM_HIP(hipMemcpy(&pos_at[p], pos.as<uint32_t>() + bounds[p], 4, hipMemcpyDeviceToHost));
        for (uint32_t p = 0; p < P; ++p)
            stride = std::max(stride, pos_at[p + 1] - pos_at[p]);
        stride = (stride + 3u) & ~3u; // float4-aligned slots
        GM_CHECK((uint64_t)P * stride < (1ull << 32), GM_ERR_RANGE, "gm_page_rank_multi: exchange vector exceeds u32");
        GM_HIP(hipMemcpy(d_bounds.p, bounds.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mg_node_map_kernel, dim3(mg_grid(n)), dim3(256), 0, 0, flag.as<uint32_t>(), pos.as<uint32_t>(),
                           d_bounds.as<uint32_t>(), P, stride, n, node_map.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
    }
    const uint64_t x_len = (uint64_t)P * stride;
    const int engine = [] {
        const char *v = getenv("GM_MULTI_ENGINE"); // tests: "pb" / "pull"; default: by the size of each slice
        return v && v[0] == 'p' && v[1] == 'b' ? GM_PR_ENGINE_PB : v && v[0] == 'p' ? GM_PR_ENGINE_PULL : GM_PR_ENGINE_AUTO;
    }();

    // ---- one rank per device: row slice, exchange buffers, engine ------------------------------------------
    std::vector<std::unique_ptr<Rank>> ranks;
    for (uint32_t p = 0; p < P; ++p) {
        ranks.emplace_back(new Rank());
        Rank &r = *ranks.back();
        r.device = devs[p];
        r.lo = bounds[p], r.hi = bounds[p + 1];
        r.send_count = pos_at[p + 1] - pos_at[p];
        const uint32_t rows = r.hi - r.lo;
        const uint64_t e0 = off_host[r.lo], e1 = off_host[r.hi], cnt = e1 - e0;
        DeviceGuard g(r.device);
        GM_HIP(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
        DevBuf d_off, d_tgt, d_map;
        GM_TRY(d_off.alloc(((size_t)rows + 1) * 4));
        GM_TRY(d_tgt.alloc((size_t)cnt * 4));
        GM_TRY(r.outdeg.alloc((size_t)rows * 4));
        GM_TRY(r.send_rows.alloc((size_t)r.send_count * 4));
        // the slice's raw arrays, then offsets rebased / targets rewritten on the rank's own device
        GM_HIP(mg_copy(d_off.p, r.device, in_csr->offsets + r.lo, src_dev, ((size_t)rows + 1) * 4));
        GM_HIP(mg_copy(d_tgt.p, r.device, in_csr->targets + e0, src_dev, (size_t)cnt * 4));
        const uint32_t *map_here = node_map.as<uint32_t>();
        if (r.device != src_dev) {
            GM_TRY(d_map.alloc((size_t)n * 4));
            GM_HIP(mg_copy(d_map.p, r.device, node_map.p, src_dev, (size_t)n * 4));
            map_here = d_map.as<uint32_t>();
        }
        {   // out-degrees and the send list are cut on the source device, then moved
            DeviceGuard gs(src_dev);
            DevBuf od, sr;
            GM_TRY(od.alloc((size_t)rows * 4));
            GM_TRY(sr.alloc((size_t)r.send_count * 4));
            if (rows)
                hipLaunchKernelGGL(mg_out_degree_kernel, dim3(mg_grid(rows)), dim3(256), 0, 0, out_csr->offsets, r.lo, rows,
                                   od.as<uint32_t>());
            if (rows)
                hipLaunchKernelGGL(mg_send_rows_kernel, dim3(mg_grid(rows)), dim3(256), 0, 0, flag.as<uint32_t>(),
                                   pos.as<uint32_t>(), r.lo, r.hi, sr.as<uint32_t>());
            GM_HIP(hipGetLastError());
            GM_HIP(hipDeviceSynchronize());
            GM_HIP(mg_copy(r.outdeg.p, r.device, od.p, src_dev, (size_t)rows * 4));
            GM_HIP(mg_copy(r.send_rows.p, r.device, sr.p, src_dev, (size_t)r.send_count * 4));
        }
        hipLaunchKernelGGL(mg_rebase_kernel, dim3(mg_grid((uint64_t)rows + 1)), dim3(256), 0, 0, d_off.as<uint32_t>(), rows + 1,
                           (uint32_t)e0);
        if (cnt)
            hipLaunchKernelGGL(mg_map_targets_kernel, dim3(mg_grid(cnt)), dim3(256), 0, 0, d_tgt.as<uint32_t>(), cnt, map_here);
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
        // hand the arrays to an owning handle: wrap, then let the Rank keep the buffers alive through the handle
        gm_csr *c = new (std::nothrow) gm_csr();
        GM_CHECK(c, GM_ERR_NOMEM, "gm_page_rank_multi: out of host memory");
        c->n = rows, c->m = cnt, c->device = r.device, c->owns = true;
        c->own_offsets = std::move(d_off);
        c->own_targets = std::move(d_tgt);
        c->offsets = c->own_offsets.as<uint32_t>();
        c->targets = c->own_targets.as<uint32_t>();
        r.rows = c;
        GM_TRY(r.scores.alloc((size_t)rows * 4));
        GM_TRY(r.x_loc.alloc((size_t)rows * 4));
        GM_TRY(r.x[0].alloc((size_t)x_len * 4));
        GM_TRY(r.x[1].alloc((size_t)x_len * 4));
        GM_TRY(r.x_send.alloc((size_t)stride * 4));
        GM_TRY(r.err.alloc(8));
        GM_HIP(hipMemset(r.x[0].p, 0, (size_t)x_len * 4));
        GM_HIP(hipMemset(r.x[1].p, 0, (size_t)x_len * 4));
        GM_HIP(hipMemset(r.x_send.p, 0, (size_t)stride * 4));
        GM_HIP(hipMemset(r.err.p, 0, 8));
        GM_HIP(hipDeviceSynchronize());
        GM_TRY(gm_pr_create_with(r.rows, n, r.lo, x_len, (uint64_t)r.outdeg.p, damping_factor, engine, &r.pr));
    }

    Comms comms;
    if (distinct) {
        GM_TRY(rccl_get(&comms.rc));
        comms.c.assign(P, nullptr);
        GM_NCCL(comms.rc, comms.rc->CommInitAll(comms.c.data(), (int)P, devs.data()));
    }

    // the out_scores every rank contributes, gathered into x[buf] of every rank
    auto exchange = [&](int buf) -> int {
        for (auto &rp : ranks) {
            Rank &r = *rp;
            DeviceGuard g(r.device);
            if (r.send_count)
                hipLaunchKernelGGL(mg_compact_kernel, dim3(mg_grid(r.send_count)), dim3(256), 0, r.st, r.x_loc.as<float>(),
                                   r.send_rows.as<uint32_t>(), r.send_count, r.x_send.as<float>());
            GM_HIP(hipGetLastError());
        }
        if (distinct) {
            GM_NCCL(comms.rc, comms.rc->GroupStart());
            for (uint32_t p = 0; p < P; ++p) {
                Rank &r = *ranks[p];
                GM_NCCL(comms.rc, comms.rc->AllGather(r.x_send.p, r.x[buf].p, stride, ncclFloat32, comms.c[p], r.st));
            }
            GM_NCCL(comms.rc, comms.rc->GroupEnd());
        } else { // virtual ranks on shared devices: the same data movement with copies
            for (auto &rp : ranks) {
                DeviceGuard g(rp->device);
                GM_HIP(hipStreamSynchronize(rp->st));
            }
            for (uint32_t p = 0; p < P; ++p)
                for (uint32_t q = 0; q < P; ++q) {
                    DeviceGuard g(ranks[q]->device);
                    float *dst = ranks[q]->x[buf].as<float>() + (size_t)p * stride;
                    if (ranks[q]->device == ranks[p]->device)
                        GM_HIP(hipMemcpyAsync(dst, ranks[p]->x_send.p, (size_t)stride * 4, hipMemcpyDeviceToDevice, ranks[q]->st));
                    else
                        GM_HIP(hipMemcpyPeerAsync(dst, ranks[q]->device, ranks[p]->x_send.p, ranks[p]->device,
                                                  (size_t)stride * 4, ranks[q]->st));
                }
            for (auto &rp : ranks) {
                DeviceGuard g(rp->device);
                GM_HIP(hipStreamSynchronize(rp->st));
            }
        }
        return GM_OK;
    };

    PinnedBuf herr;
    GM_TRY(herr.alloc((size_t)P * 8));
    for (auto &rp : ranks) {
        DeviceGuard g(rp->device);
        GM_TRY(gm_pr_init(rp->pr, (uint64_t)rp->scores.p, (uint64_t)rp->x_loc.p, rp->st));
    }
    GM_TRY(exchange(0));
    uint64_t iter = 0;
    double err = 0.0;
    int cur = 0;
    const bool can_stop_early = tolerance > 0.0;
    for (;;) {
        for (auto &rp : ranks) {
            DeviceGuard g(rp->device);
            GM_TRY(gm_pr_sweep(rp->pr, (uint64_t)rp->x[cur].p, (uint64_t)rp->x_loc.p, (uint64_t)rp->scores.p,
                               (uint64_t)rp->err.p, rp->st));
        }
        iter += 1;
        const bool last = iter == max_iterations;
        if (can_stop_early || last) {
            for (uint32_t p = 0; p < P; ++p) {
                DeviceGuard g(ranks[p]->device);
                GM_HIP(hipMemcpyAsync(herr.as<double>() + p, ranks[p]->err.p, 8, hipMemcpyDeviceToHost, ranks[p]->st));
            }
            for (auto &rp : ranks) {
                DeviceGuard g(rp->device);
                GM_HIP(hipStreamSynchronize(rp->st));
            }
            err = 0.0;
            for (uint32_t p = 0; p < P; ++p) // rank order: the same bits on every run
                err += herr.as<double>()[p];
            if (err < tolerance || last)
                break;
        }
        GM_TRY(exchange(1 - cur));
        cur = 1 - cur;
    }
    for (auto &rp : ranks) {
        DeviceGuard g(rp->device);
        if (rp->hi > rp->lo)
            GM_HIP(hipMemcpyAsync(scores_out + rp->lo, rp->scores.p, (size_t)(rp->hi - rp->lo) * 4, hipMemcpyDeviceToHost, rp->st));
    }
    for (auto &rp : ranks) {
        DeviceGuard g(rp->device);
        GM_HIP(hipStreamSynchronize(rp->st));
    }
    *iterations_out = iter;
    *error_out = err;
    return GM_OK;
}
