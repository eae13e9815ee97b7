// csr.hip — device-resident CSR handles, device-side CSR construction and relabel-by-degree,
// plus the R-MAT input generator.  (Data layer either side of the algorithm kernels.)
//
// HBM layout of a gm_csr: offsets u32[n+1], targets u32[m], weights f32[m] (SoA; the reference's
// AoS Target{target,value}, crates/builder/src/graph/mod.rs:5-10, is split so the unweighted
// algorithms stream 4 B per edge).
#include "common.hpp"
#include "device_utils.hpp"

#include <rocprim/rocprim.hpp>

#include <utility>
#include <vector>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace gm {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace gm

GM_API const char *gm_last_error(void) { return gm::g_err; }
GM_API int gm_abi_version(void) { return GM_ABI_VERSION; }

GM_API int gm_device_count(int *count_out)
{
    GM_CHECK(count_out, GM_ERR_INVALID, "gm_device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count_out = c;
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
namespace {

using namespace gm;

int new_owned_csr(uint64_t n, uint64_t m, bool weighted, int device, gm_csr **out)
{
    gm::warm_code_objects(); // (once per process)
    gm_csr *c = new (std::nothrow) gm_csr();
    GM_CHECK(c, GM_ERR_NOMEM, "out of host memory");
    c->n = n;
    c->m = m;
    c->device = device;
    c->owns = true;
    int rc;
    if ((rc = c->own_offsets.alloc((n + 1) * 4)) || (rc = c->own_targets.alloc(m * 4)) ||
        (weighted && (rc = c->own_weights.alloc(m * 4)))) {
        delete c;
        return rc;
    }
    c->offsets = c->own_offsets.as<uint32_t>();
    c->targets = c->own_targets.as<uint32_t>();
    c->weights = weighted ? c->own_weights.as<float>() : nullptr;
    *out = c;
    return GM_OK;
}

// deletes the handle under construction on every early return
struct CsrHolder {
    gm_csr *c = nullptr;
    ~CsrHolder() { delete c; }
    gm_csr *release()
    {
        gm_csr *t = c;
        c = nullptr;
        return t;
    }
};

int check_device(int device)
{
    int count = 0;
    GM_HIP(hipGetDeviceCount(&count));
    GM_CHECK(device >= 0 && device < count, GM_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    return GM_OK;
}

} // namespace

namespace {

// bit 0: offsets not monotone / beyond m; bit 1: a target >= n
__global__ void csr_validate_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint64_t n, uint64_t m,
                                    uint32_t *__restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (off[i] > off[i + 1] || off[i + 1] > m)
            b |= 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        if (tgt[i] >= n)
            b |= 2u;
    if (b)
        atomicOr(bad, b);
}

// every kernel indexes x / dist / parent by target and walks [off[u], off[u+1]): a CSR that came from a
// file or a foreign buffer is checked once, on the device (one streaming pass), before anything trusts it
int validate_csr_arrays(const uint32_t *d_off, const uint32_t *d_tgt, uint64_t n, uint64_t m, const char *who)
{
    gm::DevBuf bad;
    GM_TRY(bad.alloc(4));
    GM_HIP(hipMemset(bad.p, 0, 4));
    uint64_t work = n > m ? n : m;
    unsigned grid = (unsigned)((work + 255) / 256);
    grid = grid > 8192 ? 8192 : (grid ? grid : 1);
    hipLaunchKernelGGL(csr_validate_kernel, dim3(grid), dim3(256), 0, 0, d_off, d_tgt, n, m, bad.as<uint32_t>());
    GM_HIP(hipGetLastError());
    uint32_t h = 0;
    GM_HIP(hipMemcpy(&h, bad.p, 4, hipMemcpyDeviceToHost));
    GM_CHECK(!(h & 1u), GM_ERR_INVALID, "%s: offsets are not ascending within [0, m]", who);
    GM_CHECK(!(h & 2u), GM_ERR_RANGE, "%s: a target id is >= node_count (%llu)", who, (unsigned long long)n);
    return GM_OK;
}

} // namespace

namespace gm {
void warm_code_objects()
{
    static std::once_flag once;
    std::call_once(once, [] {
        const char *v = getenv("GM_WARM");
        if (v && *v == '0')
            return;
        warm_pagerank();
        warm_pagerank_pb();
        warm_wcc();
        warm_sssp();
        warm_tc();
        warm_multi();
    });
}
} // namespace gm

GM_API int gm_csr_upload_u32(const uint32_t *offsets, const uint32_t *targets, const float *weights, uint64_t n,
                             uint64_t m, int device, gm_csr **out)
{
    GM_CHECK(offsets && out && (targets || m == 0), GM_ERR_INVALID, "gm_csr_upload_u32: null argument");
    GM_CHECK(n < (1ull << 32) && m < (1ull << 32), GM_ERR_RANGE, "gm_csr_upload_u32: n=%llu m=%llu exceed u32",
             (unsigned long long)n, (unsigned long long)m);
    GM_CHECK(offsets[0] == 0 && offsets[n] == m, GM_ERR_INVALID, "gm_csr_upload_u32: offsets[0]=%u offsets[n]=%u, m=%llu",
             offsets[0], offsets[n], (unsigned long long)m);
    GM_TRY(check_device(device));
    gm::DeviceGuard guard(device);
    gm_csr *c = nullptr;
    GM_TRY(new_owned_csr(n, m, weights != nullptr, device, &c));
    hipError_t e = hipMemcpy(c->own_offsets.p, offsets, (n + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && m)
        e = hipMemcpy(c->own_targets.p, targets, m * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && m && weights)
        e = hipMemcpy(c->own_weights.p, weights, m * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gm::set_error("gm_csr_upload_u32: %s", hipGetErrorString(e));
        delete c;
        return GM_ERR_HIP;
    }
    const int rc = validate_csr_arrays(c->offsets, c->targets, n, m, "gm_csr_upload_u32");
    if (rc != GM_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return GM_OK;
}

GM_API int gm_csr_upload_u64(const uint64_t *offsets, const uint64_t *targets, const float *weights, uint64_t n,
                             uint64_t m, int device, gm_csr **out)
{
    GM_CHECK(offsets && out && (targets || m == 0), GM_ERR_INVALID, "gm_csr_upload_u64: null argument");
    GM_CHECK(n < (1ull << 32) && m < (1ull << 32), GM_ERR_RANGE,
             "gm_csr_upload_u64: n=%llu m=%llu do not fit the u32 device id type", (unsigned long long)n,
             (unsigned long long)m);
    std::vector<uint32_t> off32, tgt32;
    try {
        off32.resize(n + 1);
        tgt32.resize(m);
    } catch (...) {
        gm::set_error("gm_csr_upload_u64: out of host memory");
        return GM_ERR_NOMEM;
    }
    for (uint64_t i = 0; i <= n; ++i) {
        GM_CHECK(offsets[i] <= m, GM_ERR_RANGE, "gm_csr_upload_u64: offsets[%llu] out of range", (unsigned long long)i);
        off32[i] = (uint32_t)offsets[i];
    }
    for (uint64_t i = 0; i < m; ++i) {
        GM_CHECK(targets[i] < n, GM_ERR_RANGE, "gm_csr_upload_u64: targets[%llu] >= n", (unsigned long long)i);
        tgt32[i] = (uint32_t)targets[i];
    }
    return gm_csr_upload_u32(off32.data(), tgt32.data(), weights, n, m, device, out);
}

GM_API int gm_csr_wrap_device(uint64_t d_offsets, uint64_t d_targets, uint64_t d_weights, uint64_t n, uint64_t m,
                              int device, gm_csr **out)
{
    GM_CHECK(d_offsets && out && (d_targets || m == 0), GM_ERR_INVALID, "gm_csr_wrap_device: null argument");
    GM_CHECK(n < (1ull << 32) && m < (1ull << 32), GM_ERR_RANGE, "gm_csr_wrap_device: n or m exceed u32");
    GM_TRY(check_device(device));
    {
        gm::DeviceGuard guard(device);
        GM_TRY(validate_csr_arrays(reinterpret_cast<const uint32_t *>(d_offsets), reinterpret_cast<const uint32_t *>(d_targets),
                                   n, m, "gm_csr_wrap_device"));
    }
    gm::warm_code_objects(); // (once per process)
    gm_csr *c = new (std::nothrow) gm_csr();
    GM_CHECK(c, GM_ERR_NOMEM, "out of host memory");
    c->n = n;
    c->m = m;
    c->device = device;
    c->offsets = reinterpret_cast<const uint32_t *>(d_offsets);
    c->targets = reinterpret_cast<const uint32_t *>(d_targets);
    c->weights = reinterpret_cast<const float *>(d_weights);
    *out = c;
    return GM_OK;
}

GM_API void gm_csr_free(gm_csr *csr)
{
    if (!csr)
        return;
    gm::DeviceGuard guard(csr->device);
    delete csr;
}

GM_API int gm_csr_set_source_flags(gm_csr *csr, uint64_t d_flags, uint64_t len)
{
    GM_CHECK(csr, GM_ERR_INVALID, "gm_csr_set_source_flags: null handle");
    gm::DeviceGuard guard(csr->device);
    std::map<uint64_t, std::shared_ptr<const gm::PbPlan>> plans; // (a plan built without the flags is not the plan with them)
    {
        std::lock_guard<std::mutex> lock(csr->cache_mu);
        plans.swap(csr->pb_plans);
    }
    plans.clear();
    csr->source_flags.release();
    csr->source_flags_len = 0;
    if (!d_flags || !len)
        return GM_OK;
    GM_TRY(csr->source_flags.alloc((size_t)len));
    GM_HIP(hipMemcpy(csr->source_flags.p, reinterpret_cast<const void *>(d_flags), (size_t)len, hipMemcpyDeviceToDevice));
    csr->source_flags_len = len;
    return GM_OK;
}

GM_API int gm_csr_trim(const gm_csr *csr)
{
    GM_CHECK(csr, GM_ERR_INVALID, "gm_csr_trim: null handle");
    gm::DeviceGuard guard(csr->device);
    // taken out under the lock, destroyed outside it (a destructor may synchronise the device)
    std::map<uint64_t, std::shared_ptr<const gm::PbPlan>> plans;
    std::unique_ptr<gm::SsspScratch> sssp;
    std::shared_ptr<const gm::SsspOrder> sssp_order;
    std::unique_ptr<gm::WccScratch> wcc;
    std::shared_ptr<gm::PrCallState> pr;
    std::shared_ptr<const gm::TcDag> dag;
    std::unique_ptr<gm::MultiState, gm::MultiStateDeleter> multi;
    {
        std::lock_guard<std::mutex> lock(csr->cache_mu);
        multi = std::move(csr->multi);
        plans.swap(csr->pb_plans);
        sssp = std::move(csr->sssp_scratch);
        sssp_order = std::move(csr->sssp_order);
        csr->sssp_order_failed.store(0, std::memory_order_relaxed); // room may have been made: the next loop of calls may build them
        wcc = std::move(csr->wcc_scratch);
        pr = std::move(csr->pr_call);
        dag = std::move(csr->tc_dag);
    }
    return GM_OK;
}

GM_API uint64_t gm_csr_node_count(const gm_csr *csr) { return csr ? csr->n : 0; }
GM_API uint64_t gm_csr_edge_count(const gm_csr *csr) { return csr ? csr->m : 0; }
GM_API int gm_csr_device(const gm_csr *csr) { return csr ? csr->device : -1; }
GM_API uint64_t gm_csr_offsets_ptr(const gm_csr *csr) { return csr ? (uint64_t)csr->offsets : 0; }
GM_API uint64_t gm_csr_targets_ptr(const gm_csr *csr) { return csr ? (uint64_t)csr->targets : 0; }
GM_API uint64_t gm_csr_weights_ptr(const gm_csr *csr) { return csr ? (uint64_t)csr->weights : 0; }

GM_API int gm_csr_download(const gm_csr *csr, uint32_t *offsets, uint32_t *targets, float *weights)
{
    GM_CHECK(csr, GM_ERR_INVALID, "gm_csr_download: null handle");
    gm::DeviceGuard guard(csr->device);
    if (offsets)
        GM_HIP(hipMemcpy(offsets, csr->offsets, (csr->n + 1) * 4, hipMemcpyDeviceToHost));
    if (targets && csr->m)
        GM_HIP(hipMemcpy(targets, csr->targets, csr->m * 4, hipMemcpyDeviceToHost));
    if (weights && csr->m) {
        GM_CHECK(csr->weights, GM_ERR_INVALID, "gm_csr_download: the CSR carries no weights");
        GM_HIP(hipMemcpy(weights, csr->weights, csr->m * 4, hipMemcpyDeviceToHost));
    }
    return GM_OK;
}

namespace {
__global__ void degrees_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t *__restrict__ deg)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride)
        deg[u] = off[u + 1] - off[u];
}
} // namespace

GM_API int gm_csr_degrees(const gm_csr *csr, uint32_t *degrees_out)
{
    GM_CHECK(csr && (degrees_out || csr->n == 0), GM_ERR_INVALID, "gm_csr_degrees: null argument");
    if (csr->n == 0)
        return GM_OK;
    gm::DeviceGuard guard(csr->device);
    gm::DevBuf d;
    GM_TRY(d.alloc(csr->n * 4));
    hipLaunchKernelGGL(degrees_kernel, dim3(gm::div_up(csr->n, 256) > 4096 ? 4096 : gm::div_up(csr->n, 256)),
                       dim3(256), 0, 0, csr->offsets, (uint32_t)csr->n, d.as<uint32_t>());
    GM_HIP(hipGetLastError());
    GM_HIP(hipMemcpy(degrees_out, d.p, csr->n * 4, hipMemcpyDeviceToHost));
    return GM_OK;
}

namespace {
constexpr uint32_t kMaxParts = 64;
struct PartBounds {
    uint32_t parts, stride;
    uint32_t b[kMaxParts + 1];
};

__global__ void slice_offsets_kernel(const uint32_t *__restrict__ off, uint32_t row_lo, uint32_t rows,
                                     uint32_t *__restrict__ new_off)
{
    const uint32_t base = off[row_lo];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= rows; i += stride)
        new_off[i] = off[row_lo + i] - base;
}

__global__ void slice_targets_kernel(const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint64_t first,
                                     uint64_t count, PartBounds pb, uint32_t *__restrict__ new_tgt,
                                     float *__restrict__ new_w)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint32_t v = tgt[first + i];
        if (pb.parts) {
            uint32_t p = 0;
            while (p + 1 < pb.parts && v >= pb.b[p + 1])
                ++p;
            v = p * pb.stride + (v - pb.b[p]);
        }
        new_tgt[i] = v;
        if (new_w)
            new_w[i] = w[first + i];
    }
}
} // namespace

GM_API int gm_csr_slice_rows(const gm_csr *full, uint64_t row_lo, uint64_t row_hi, const uint32_t *bounds,
                             uint32_t parts, uint32_t stride, gm_csr **out)
{
    GM_CHECK(full && out, GM_ERR_INVALID, "gm_csr_slice_rows: null argument");
    GM_CHECK(row_lo <= row_hi && row_hi <= full->n, GM_ERR_RANGE, "gm_csr_slice_rows: rows [%llu, %llu) outside [0, %llu)",
             (unsigned long long)row_lo, (unsigned long long)row_hi, (unsigned long long)full->n);
    PartBounds pb;
    pb.parts = 0;
    pb.stride = stride;
    if (bounds) {
        GM_CHECK(parts >= 1 && parts <= kMaxParts, GM_ERR_RANGE, "gm_csr_slice_rows: parts %u not in [1, %u]", parts, kMaxParts);
        GM_CHECK(bounds[0] == 0 && bounds[parts] == full->n, GM_ERR_INVALID, "gm_csr_slice_rows: bounds must span [0, n]");
        for (uint32_t p = 0; p < parts; ++p) {
            GM_CHECK(bounds[p] <= bounds[p + 1] && bounds[p + 1] - bounds[p] <= stride, GM_ERR_INVALID,
                     "gm_csr_slice_rows: part %u has %u rows > stride %u (or bounds not ascending)", p,
                     bounds[p + 1] - bounds[p], stride);
            GM_CHECK((uint64_t)(p + 1) * stride <= (1ull << 32), GM_ERR_RANGE, "gm_csr_slice_rows: padded index space exceeds u32");
        }
        pb.parts = parts;
        memcpy(pb.b, bounds, (parts + 1) * sizeof(uint32_t));
    }
    gm::DeviceGuard guard(full->device);
    uint32_t lohi[2] = {0, 0};
    GM_HIP(hipMemcpy(&lohi[0], full->offsets + row_lo, 4, hipMemcpyDeviceToHost));
    GM_HIP(hipMemcpy(&lohi[1], full->offsets + row_hi, 4, hipMemcpyDeviceToHost));
    const uint64_t rows = row_hi - row_lo, count = lohi[1] - lohi[0];
    gm_csr *c = nullptr;
    GM_TRY(new_owned_csr(rows, count, full->weights != nullptr, full->device, &c));
    hipLaunchKernelGGL(slice_offsets_kernel, dim3(gm::div_up(rows + 1, 256) > 8192 ? 8192 : gm::div_up(rows + 1, 256)),
                       dim3(256), 0, 0, full->offsets, (uint32_t)row_lo, (uint32_t)rows, c->own_offsets.as<uint32_t>());
    if (count)
        hipLaunchKernelGGL(slice_targets_kernel, dim3(gm::div_up(count, 256) > 8192 ? 8192 : gm::div_up(count, 256)),
                           dim3(256), 0, 0, full->targets, full->weights, (uint64_t)lohi[0], count, pb,
                           c->own_targets.as<uint32_t>(), full->weights ? c->own_weights.as<float>() : nullptr);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        gm::set_error("gm_csr_slice_rows: kernel failure");
        delete c;
        return GM_ERR_HIP;
    }
    *out = c;
    return GM_OK;
}

namespace {
__global__ void slice_targets_map_kernel(const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint64_t first,
                                         uint64_t count, const uint32_t *__restrict__ map, uint32_t *__restrict__ new_tgt,
                                         float *__restrict__ new_w)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        new_tgt[i] = map[tgt[first + i]];
        if (new_w)
            new_w[i] = w[first + i];
    }
}
} // namespace

GM_API int gm_csr_slice_rows_map(const gm_csr *full, uint64_t row_lo, uint64_t row_hi, uint64_t d_map, gm_csr **out)
{
    GM_CHECK(full && out && d_map, GM_ERR_INVALID, "gm_csr_slice_rows_map: null argument");
    GM_CHECK(row_lo <= row_hi && row_hi <= full->n, GM_ERR_RANGE, "gm_csr_slice_rows_map: rows [%llu, %llu) outside [0, %llu)",
             (unsigned long long)row_lo, (unsigned long long)row_hi, (unsigned long long)full->n);
    gm::DeviceGuard guard(full->device);
    uint32_t lohi[2] = {0, 0};
    GM_HIP(hipMemcpy(&lohi[0], full->offsets + row_lo, 4, hipMemcpyDeviceToHost));
    GM_HIP(hipMemcpy(&lohi[1], full->offsets + row_hi, 4, hipMemcpyDeviceToHost));
    const uint64_t rows = row_hi - row_lo, count = lohi[1] - lohi[0];
    gm_csr *c = nullptr;
    GM_TRY(new_owned_csr(rows, count, full->weights != nullptr, full->device, &c));
    hipLaunchKernelGGL(slice_offsets_kernel, dim3(gm::div_up(rows + 1, 256) > 8192 ? 8192 : gm::div_up(rows + 1, 256)),
                       dim3(256), 0, 0, full->offsets, (uint32_t)row_lo, (uint32_t)rows, c->own_offsets.as<uint32_t>());
    if (count)
        hipLaunchKernelGGL(slice_targets_map_kernel, dim3(gm::div_up(count, 256) > 8192 ? 8192 : gm::div_up(count, 256)),
                           dim3(256), 0, 0, full->targets, full->weights, (uint64_t)lohi[0], count,
                           reinterpret_cast<const uint32_t *>(d_map), c->own_targets.as<uint32_t>(),
                           full->weights ? c->own_weights.as<float>() : nullptr);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        gm::set_error("gm_csr_slice_rows_map: kernel failure");
        delete c;
        return GM_ERR_HIP;
    }
    *out = c;
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// R-MAT generator — integer-only, identical to oracle/graph_oracle.c:orc_rmat_edge
// ------------------------------------------------------------------------------------------------
namespace {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr uint32_t RMAT_T_A = 2448131358u;   // floor(0.57 * 2^32)
constexpr uint32_t RMAT_T_AB = 3264175145u;  // floor(0.76 * 2^32)
constexpr uint32_t RMAT_T_ABC = 4080218931u; // floor(0.95 * 2^32)

struct Scrambler {
    uint32_t mask, k1, k2, k3, sh;
    __device__ __forceinline__ uint32_t operator()(uint32_t x) const
    {
        x = (x * k1) & mask;
        x ^= x >> sh;
        x = (x * k2) & mask;
        x ^= x >> sh;
        x = (x + k3) & mask;
        return x;
    }
};

__global__ void rmat_edges_kernel(uint32_t scale, uint64_t seed, uint64_t first, uint64_t count, Scrambler sc,
                                  uint32_t *__restrict__ src, uint32_t *__restrict__ dst)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t idx = first + i;
        uint32_t s = 0, d = 0;
        for (uint32_t lvl = 0; lvl < scale; lvl += 2) {
            const uint64_t h = splitmix64(seed ^ (idx * 32u + (lvl >> 1)) * 0xD1342543DE82EF95ull);
            const uint32_t r0 = (uint32_t)h, r1 = (uint32_t)(h >> 32);
            uint32_t sb = r0 >= RMAT_T_AB;
            uint32_t db = (r0 >= RMAT_T_A && r0 < RMAT_T_AB) || r0 >= RMAT_T_ABC;
            s = (s << 1) | sb;
            d = (d << 1) | db;
            if (lvl + 1 < scale) {
                sb = r1 >= RMAT_T_AB;
                db = (r1 >= RMAT_T_A && r1 < RMAT_T_AB) || r1 >= RMAT_T_ABC;
                s = (s << 1) | sb;
                d = (d << 1) | db;
            }
        }
        src[i] = sc(s);
        dst[i] = sc(d);
    }
}

__global__ void rmat_weights_kernel(uint64_t seed, uint64_t first, uint64_t count, float *__restrict__ w)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t h = splitmix64((seed ^ 0xA0761D6478BD642Full) + (first + i) * 0xE7037ED1A0B428DBull);
        w[i] = (float)((uint32_t)(h >> 40) + 1u) * (1.0f / 16777216.0f);
    }
}

unsigned stream_grid(uint64_t count)
{
    unsigned g = gm::div_up(count, 256);
    return g > 256 * 32 ? 256 * 32 : (g ? g : 1);
}

} // namespace

GM_API int gm_rmat_edges_device(uint32_t scale, uint64_t seed, uint64_t first_edge, uint64_t count, uint64_t d_src,
                                uint64_t d_dst, int device, void *stream)
{
    GM_CHECK(scale >= 1 && scale <= 31, GM_ERR_INVALID, "gm_rmat_edges_device: scale %u not in [1, 31]", scale);
    GM_CHECK((d_src && d_dst) || count == 0, GM_ERR_INVALID, "gm_rmat_edges_device: null buffer");
    if (count == 0)
        return GM_OK;
    gm::DeviceGuard guard(device);
    const uint64_t sseed = seed ^ 0x5851F42D4C957F2Dull;
    Scrambler sc;
    sc.mask = (1u << scale) - 1u;
    sc.k1 = (uint32_t)(splitmix64(sseed) | 1u);
    sc.k2 = (uint32_t)(splitmix64(sseed + 1) | 1u);
    sc.k3 = (uint32_t)splitmix64(sseed + 2);
    sc.sh = (scale + 1) / 2;
    hipLaunchKernelGGL(rmat_edges_kernel, dim3(stream_grid(count)), dim3(256), 0, (hipStream_t)stream, scale, seed,
                       first_edge, count, sc, reinterpret_cast<uint32_t *>(d_src), reinterpret_cast<uint32_t *>(d_dst));
    GM_HIP(hipGetLastError());
    return GM_OK;
}

GM_API int gm_rmat_weights_device(uint64_t seed, uint64_t first_edge, uint64_t count, uint64_t d_weights, int device,
                                  void *stream)
{
    GM_CHECK(d_weights || count == 0, GM_ERR_INVALID, "gm_rmat_weights_device: null buffer");
    if (count == 0)
        return GM_OK;
    gm::DeviceGuard guard(device);
    hipLaunchKernelGGL(rmat_weights_kernel, dim3(stream_grid(count)), dim3(256), 0, (hipStream_t)stream, seed,
                       first_edge, count, reinterpret_cast<float *>(d_weights));
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-side CSR construction (crates/builder/src/graph/csr.rs:124-221, :886-948)
//   entries e in [0, total): for Undirected the first m entries are the out-direction pass
//   (row = src, col = dst), the next m the in-direction pass (row = dst, col = src) — the
//   reference's scatter order (csr.rs:154-172).  A stable LSD radix sort on the row (Unsorted)
//   or on row<<32|col (Sorted / Deduplicated) reproduces the sequential arrival order.
// ------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ void entry_row_col(uint64_t e, uint64_t m, int direction, const uint32_t *__restrict__ src,
                                              const uint32_t *__restrict__ dst, uint32_t &row, uint32_t &col)
{
    if (direction == GM_DIR_OUTGOING || (direction == GM_DIR_UNDIRECTED && e < m)) {
        row = src[e];
        col = dst[e];
    } else {
        const uint64_t i = direction == GM_DIR_UNDIRECTED ? e - m : e;
        row = dst[i];
        col = src[i];
    }
}

__global__ void make_keys64_kernel(uint64_t total, uint64_t m, int direction, const uint32_t *__restrict__ src,
                                   const uint32_t *__restrict__ dst, uint64_t *__restrict__ keys,
                                   uint32_t *__restrict__ idx_or_null, uint32_t n, uint32_t *__restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        uint32_t r, c;
        entry_row_col(e, m, direction, src, dst, r, c);
        if (r >= n || c >= n)
            *bad = 1;
        keys[e] = ((uint64_t)r << 32) | c;
        if (idx_or_null)
            idx_or_null[e] = (uint32_t)e;
    }
}

__global__ void make_keys32_kernel(uint64_t total, uint64_t m, int direction, const uint32_t *__restrict__ src,
                                   const uint32_t *__restrict__ dst, uint32_t *__restrict__ keys,
                                   uint32_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        uint32_t r, c;
        entry_row_col(e, m, direction, src, dst, r, c);
        if (r >= n || c >= n)
            *bad = 1;
        keys[e] = r;
        idx[e] = (uint32_t)e;
    }
}

// offsets from sorted row keys, no atomics: position i opens every row in (row[i-1], row[i]]
template <class KeyT, int SHIFT>
__global__ void offsets_from_sorted_kernel(const KeyT *__restrict__ keys, uint64_t total, uint32_t n,
                                           uint32_t *__restrict__ off)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= total; i += stride) {
        const uint32_t lo = i == 0 ? 0u : (uint32_t)(keys[i - 1] >> SHIFT) + 1u;
        const uint32_t hi = i == total ? n : (uint32_t)(keys[i] >> SHIFT);
        for (uint32_t r = lo; r <= hi; ++r)
            off[r] = (uint32_t)i;
    }
}

__global__ void targets_from_keys64_kernel(const uint64_t *__restrict__ keys, uint64_t total, uint32_t *__restrict__ tgt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        tgt[i] = (uint32_t)keys[i];
}

__global__ void gather_by_idx_kernel(const uint32_t *__restrict__ idx, uint64_t total, uint64_t m, int direction,
                                     const uint32_t *__restrict__ src, const uint32_t *__restrict__ dst,
                                     const float *__restrict__ w, uint32_t *__restrict__ tgt_or_null,
                                     float *__restrict__ w_out_or_null)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t e = idx[i];
        if (tgt_or_null) {
            uint32_t r, c;
            entry_row_col(e, m, direction, src, dst, r, c);
            tgt_or_null[i] = c;
        }
        if (w_out_or_null)
            w_out_or_null[i] = w[(direction == GM_DIR_UNDIRECTED && e >= m) ? e - m : e];
    }
}

// Deduplicated: keep the first of each run of equal (row, col), drop row == col
__global__ void dedup_flags_kernel(const uint64_t *__restrict__ keys, uint64_t total, uint32_t *__restrict__ keep)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t k = keys[i];
        const bool first = i == 0 || keys[i - 1] != k;
        keep[i] = (first && (uint32_t)(k >> 32) != (uint32_t)k) ? 1u : 0u;
    }
}

__global__ void dedup_scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ keep,
                                     const uint32_t *__restrict__ pos, uint64_t total,
                                     const float *__restrict__ w_sorted_or_null, uint32_t *__restrict__ tgt,
                                     float *__restrict__ w_out_or_null)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
        if (keep[i]) {
            tgt[pos[i]] = (uint32_t)keys[i];
            if (w_out_or_null)
                w_out_or_null[pos[i]] = w_sorted_or_null[i];
        }
}

__global__ void remap_offsets_kernel(const uint32_t *__restrict__ old_off, const uint32_t *__restrict__ pos,
                                     uint32_t n, uint64_t total, uint32_t new_total, uint32_t *__restrict__ new_off)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride) {
        const uint32_t o = old_off[r];
        new_off[r] = o >= total ? new_total : pos[o];
    }
}

int ceil_log2(uint64_t x)
{
    int b = 0;
    while ((1ull << b) < x)
        ++b;
    return b < 1 ? 1 : b;
}

template <class K> int radix_sort_keys_inplace(gm::DevBuf &keys, gm::DevBuf &alt, uint64_t count, int begin_bit, int end_bit)
{
    rocprim::double_buffer<K> db(keys.as<K>(), alt.as<K>());
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, db, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    gm::DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, db, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    GM_HIP(hipDeviceSynchronize());
    if (db.current() != keys.as<K>())
        std::swap(keys, alt);
    return GM_OK;
}

template <class K>
int radix_sort_pairs_inplace(gm::DevBuf &keys, gm::DevBuf &kalt, gm::DevBuf &vals, gm::DevBuf &valt, uint64_t count,
                             int begin_bit, int end_bit)
{
    rocprim::double_buffer<K> dk(keys.as<K>(), kalt.as<K>());
    rocprim::double_buffer<uint32_t> dv(vals.as<uint32_t>(), valt.as<uint32_t>());
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    gm::DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    GM_HIP(hipDeviceSynchronize());
    if (dk.current() != keys.as<K>())
        std::swap(keys, kalt);
    if (dv.current() != vals.as<uint32_t>())
        std::swap(vals, valt);
    return GM_OK;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, uint64_t count)
{
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, count, rocprim::plus<uint32_t>(), (hipStream_t)0));
    gm::DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, 0u, count, rocprim::plus<uint32_t>(), (hipStream_t)0));
    return GM_OK;
}

} // namespace

static int gm_csr_build_device_impl(uint64_t n, uint64_t m, uint64_t d_src, uint64_t d_dst, uint64_t d_weights,
                               int direction, int layout, int device, gm_csr **out)
{
    GM_CHECK(out && ((d_src && d_dst) || m == 0), GM_ERR_INVALID, "gm_csr_build_device: null argument");
    GM_CHECK(direction >= GM_DIR_OUTGOING && direction <= GM_DIR_UNDIRECTED, GM_ERR_INVALID, "bad direction %d", direction);
    GM_CHECK(layout >= GM_LAYOUT_UNSORTED && layout <= GM_LAYOUT_DEDUPLICATED, GM_ERR_INVALID, "bad layout %d", layout);
    const uint64_t total = direction == GM_DIR_UNDIRECTED ? 2 * m : m;
    GM_CHECK(n < (1ull << 32) && total < (1ull << 32), GM_ERR_RANGE, "gm_csr_build_device: n=%llu entries=%llu exceed u32",
             (unsigned long long)n, (unsigned long long)total);
    GM_TRY(check_device(device));
    gm::DeviceGuard guard(device);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(d_src);
    const uint32_t *dst = reinterpret_cast<const uint32_t *>(d_dst);
    const float *w = reinterpret_cast<const float *>(d_weights);
    const bool weighted = w != nullptr;
    const unsigned grid = stream_grid(total);
    const int row_bits = ceil_log2(n ? n : 1);

    gm::DevBuf bad;
    GM_TRY(bad.alloc(4));
    GM_HIP(hipMemset(bad.p, 0, 4));

    CsrHolder hold;
    gm_csr *&c = hold.c;
    if (total == 0) {
        GM_TRY(new_owned_csr(n, 0, weighted, device, &c));
        GM_HIP(hipMemset(c->own_offsets.p, 0, (n + 1) * 4));
        *out = hold.release();
        return GM_OK;
    }

    auto fail = [&](int rc) { return rc; }; // `hold` frees the handle
    // An endpoint >= node_count must stop the build BEFORE the sort: the radix sort covers only
    // ceil_log2(n) row bits, so an oversized row id would survive in the keys and send
    // offsets_from_sorted_kernel far past the n + 1 offsets (heap corruption / a near-endless loop).
    auto endpoints_ok = [&]() -> int {
        uint32_t hbad = 0;
        GM_HIP(hipMemcpy(&hbad, bad.p, 4, hipMemcpyDeviceToHost)); // orders after the key kernel (null stream)
        GM_CHECK(!hbad, GM_ERR_RANGE, "gm_csr_build_device: an edge endpoint is >= node_count (%llu)", (unsigned long long)n);
        return GM_OK;
    };

    if (layout == GM_LAYOUT_UNSORTED) {
        gm::DevBuf keys, kalt, idx, ialt;
        GM_TRY(keys.alloc_big(total * 4));
        GM_TRY(kalt.alloc_big(total * 4));
        GM_TRY(idx.alloc_big(total * 4));
        GM_TRY(ialt.alloc_big(total * 4));
        hipLaunchKernelGGL(make_keys32_kernel, dim3(grid), dim3(256), 0, 0, total, m, direction, src, dst,
                           keys.as<uint32_t>(), idx.as<uint32_t>(), (uint32_t)n, bad.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_TRY(endpoints_ok());
        GM_TRY(radix_sort_pairs_inplace<uint32_t>(keys, kalt, idx, ialt, total, 0, row_bits));
        GM_TRY(new_owned_csr(n, total, weighted, device, &c));
        hipLaunchKernelGGL((offsets_from_sorted_kernel<uint32_t, 0>), dim3(grid), dim3(256), 0, 0, keys.as<uint32_t>(),
                           total, (uint32_t)n, c->own_offsets.as<uint32_t>());
        hipLaunchKernelGGL(gather_by_idx_kernel, dim3(grid), dim3(256), 0, 0, idx.as<uint32_t>(), total, m, direction,
                           src, dst, w, c->own_targets.as<uint32_t>(), weighted ? c->own_weights.as<float>() : nullptr);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            gm::set_error("gm_csr_build_device: kernel failure");
            return fail(GM_ERR_HIP);
        }
    } else {
        gm::DevBuf keys, kalt, idx, ialt, wsorted;
        GM_TRY(keys.alloc_big(total * 8));
        GM_TRY(kalt.alloc_big(total * 8));
        if (weighted) {
            GM_TRY(idx.alloc_big(total * 4));
            GM_TRY(ialt.alloc_big(total * 4));
        }
        hipLaunchKernelGGL(make_keys64_kernel, dim3(grid), dim3(256), 0, 0, total, m, direction, src, dst,
                           keys.as<uint64_t>(), weighted ? idx.as<uint32_t>() : nullptr, (uint32_t)n, bad.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_TRY(endpoints_ok());
        if (weighted)
            GM_TRY(radix_sort_pairs_inplace<uint64_t>(keys, kalt, idx, ialt, total, 0, 32 + row_bits));
        else
            GM_TRY(radix_sort_keys_inplace<uint64_t>(keys, kalt, total, 0, 32 + row_bits));
        kalt.release();
        ialt.release();

        if (layout == GM_LAYOUT_SORTED) {
            GM_TRY(new_owned_csr(n, total, weighted, device, &c));
            hipLaunchKernelGGL((offsets_from_sorted_kernel<uint64_t, 32>), dim3(grid), dim3(256), 0, 0,
                               keys.as<uint64_t>(), total, (uint32_t)n, c->own_offsets.as<uint32_t>());
            hipLaunchKernelGGL(targets_from_keys64_kernel, dim3(grid), dim3(256), 0, 0, keys.as<uint64_t>(), total,
                               c->own_targets.as<uint32_t>());
            if (weighted)
                hipLaunchKernelGGL(gather_by_idx_kernel, dim3(grid), dim3(256), 0, 0, idx.as<uint32_t>(), total, m,
                                   direction, src, dst, w, (uint32_t *)nullptr, c->own_weights.as<float>());
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                gm::set_error("gm_csr_build_device: kernel failure");
                return fail(GM_ERR_HIP);
            }
        } else { // Deduplicated
            gm::DevBuf keep, pos, old_off;
            GM_TRY(keep.alloc_big((total + 1) * 4));
            GM_TRY(pos.alloc_big((total + 1) * 4));
            GM_TRY(old_off.alloc((n + 1) * 4));
            if (weighted) {
                GM_TRY(wsorted.alloc_big(total * 4));
                hipLaunchKernelGGL(gather_by_idx_kernel, dim3(grid), dim3(256), 0, 0, idx.as<uint32_t>(), total, m,
                                   direction, src, dst, w, (uint32_t *)nullptr, wsorted.as<float>());
            }
            hipLaunchKernelGGL((offsets_from_sorted_kernel<uint64_t, 32>), dim3(grid), dim3(256), 0, 0,
                               keys.as<uint64_t>(), total, (uint32_t)n, old_off.as<uint32_t>());
            hipLaunchKernelGGL(dedup_flags_kernel, dim3(grid), dim3(256), 0, 0, keys.as<uint64_t>(), total,
                               keep.as<uint32_t>());
            GM_HIP(hipMemset(keep.as<uint32_t>() + total, 0, 4));
            GM_HIP(hipGetLastError());
            GM_TRY(exclusive_scan_u32(keep.as<uint32_t>(), pos.as<uint32_t>(), total + 1));
            uint32_t new_total = 0;
            GM_HIP(hipMemcpy(&new_total, pos.as<uint32_t>() + total, 4, hipMemcpyDeviceToHost));
            GM_TRY(new_owned_csr(n, new_total, weighted, device, &c));
            hipLaunchKernelGGL(dedup_scatter_kernel, dim3(grid), dim3(256), 0, 0, keys.as<uint64_t>(), keep.as<uint32_t>(),
                               pos.as<uint32_t>(), total, weighted ? wsorted.as<float>() : nullptr,
                               c->own_targets.as<uint32_t>(), weighted ? c->own_weights.as<float>() : nullptr);
            hipLaunchKernelGGL(remap_offsets_kernel, dim3(stream_grid(n + 1)), dim3(256), 0, 0, old_off.as<uint32_t>(),
                               pos.as<uint32_t>(), (uint32_t)n, total, new_total, c->own_offsets.as<uint32_t>());
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                gm::set_error("gm_csr_build_device: kernel failure");
                return fail(GM_ERR_HIP);
            }
        }
    }
    *out = hold.release();
    return GM_OK;
}

GM_API int gm_csr_build_device(uint64_t n, uint64_t m, uint64_t d_src, uint64_t d_dst, uint64_t d_weights, int direction,
                               int layout, int device, gm_csr **out)
{
    struct Site {
        int prev;
        Site() : prev(gm::arena_site()) { gm::arena_site() = 1; }
        ~Site() { gm::arena_site() = prev; }
    } site;
    return gm_csr_build_device_impl(n, m, d_src, d_dst, d_weights, direction, layout, device, out);
}

GM_API int gm_csr_build_host(uint64_t n, uint64_t m, const uint32_t *src, const uint32_t *dst, const float *weights,
                             int direction, int layout, int device, gm_csr **out)
{
    GM_CHECK(out && ((src && dst) || m == 0), GM_ERR_INVALID, "gm_csr_build_host: null argument");
    GM_TRY(check_device(device));
    gm::DeviceGuard guard(device);
    gm::DevBuf ds, dd, dw;
    GM_TRY(ds.alloc(m * 4));
    GM_TRY(dd.alloc(m * 4));
    if (m) {
        GM_HIP(hipMemcpy(ds.p, src, m * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemcpy(dd.p, dst, m * 4, hipMemcpyHostToDevice));
    }
    if (weights) {
        GM_TRY(dw.alloc(m * 4));
        if (m)
            GM_HIP(hipMemcpy(dw.p, weights, m * 4, hipMemcpyHostToDevice));
    }
    return gm_csr_build_device(n, m, (uint64_t)ds.p, (uint64_t)dd.p, weights ? (uint64_t)dw.p : 0, direction, layout,
                               device, out);
}

// ------------------------------------------------------------------------------------------------
// to_undirected (crates/builder/src/graph_ops.rs:176-230, csr.rs:391-464): an Undirected build over the
// out-edges of a directed graph, entirely on the device.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void expand_rows_kernel(const uint32_t *__restrict__ off, uint32_t n,
                                                          uint32_t *__restrict__ src)
{
    const uint32_t lane = threadIdx.x & (gm::kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + gm::kWave - 1) / gm::kWave * gm::kWave;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_pad; r += stride) {
        uint32_t s = 0, e = 0;
        if (r < n) {
            s = off[r];
            e = off[r + 1];
        }
        const uint32_t len = e - s;
        if (len <= 32)
            for (uint32_t i = s; i < e; ++i)
                src[i] = r;
        uint64_t big = __ballot(len > 32);
        while (big) {
            const int from = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t br = __shfl(r, from, gm::kWave), bs = __shfl(s, from, gm::kWave), be = __shfl(e, from, gm::kWave);
            for (uint32_t i = bs + lane; i < be; i += gm::kWave)
                src[i] = br;
        }
    }
}
} // namespace

GM_API int gm_csr_to_undirected(const gm_csr *out_csr, int layout, gm_csr **out)
{
    GM_CHECK(out_csr && out, GM_ERR_INVALID, "gm_csr_to_undirected: null argument");
    gm::DeviceGuard guard(out_csr->device);
    gm::DevBuf src;
    GM_TRY(src.alloc(out_csr->m * 4));
    if (out_csr->m) {
        hipLaunchKernelGGL(expand_rows_kernel, dim3(stream_grid(out_csr->n)), dim3(256), 0, 0, out_csr->offsets,
                           (uint32_t)out_csr->n, src.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
    }
    return gm_csr_build_device(out_csr->n, out_csr->m, (uint64_t)src.p, (uint64_t)out_csr->targets,
                               (uint64_t)out_csr->weights, GM_DIR_UNDIRECTED, layout, out_csr->device, out);
}

// ------------------------------------------------------------------------------------------------
// make_degree_ordered (crates/builder/src/graph_ops.rs:511-638) on the device
// ------------------------------------------------------------------------------------------------
namespace {

__global__ void degree_pairs_kernel(const uint32_t *__restrict__ off, uint32_t n, uint64_t *__restrict__ pairs)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride)
        pairs[u] = ((uint64_t)(off[u + 1] - off[u]) << 32) | u;
}

// pairs sorted descending: rank k holds (degree, old node)
__global__ void unzip_pairs_kernel(const uint64_t *__restrict__ pairs, uint32_t n, uint32_t *__restrict__ new_id,
                                   uint32_t *__restrict__ new_deg)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k <= n; k += stride) {
        if (k == n) {
            new_deg[k] = 0;
            continue;
        }
        const uint64_t p = pairs[k];
        new_id[(uint32_t)p] = k;
        new_deg[k] = (uint32_t)(p >> 32);
    }
}

__global__ void relabel_keys_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt, uint32_t n,
                                    uint64_t m, const uint32_t *__restrict__ new_id, uint64_t *__restrict__ keys)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const uint32_t u = gm::row_of_entry(off, n, (uint32_t)e);
        keys[e] = ((uint64_t)new_id[u] << 32) | new_id[tgt[e]];
    }
}

} // namespace

GM_API int gm_csr_relabel_by_degree(const gm_csr *g, gm_csr **out, uint32_t *new_id_out)
{
    GM_CHECK(g && out, GM_ERR_INVALID, "gm_csr_relabel_by_degree: null argument");
    GM_CHECK(!g->weights, GM_ERR_UNSUPPORTED,
             "gm_csr_relabel_by_degree: weighted graphs are not relabelled (reference requires EV: Ord)");
    gm::DeviceGuard guard(g->device);
    const uint64_t n = g->n, m = g->m;
    CsrHolder hold;
    gm_csr *&c = hold.c;
    GM_TRY(new_owned_csr(n, m, false, g->device, &c));
    auto fail = [&](int rc) { return rc; }; // `hold` frees the handle
    if (n == 0) {
        GM_HIP(hipMemset(c->own_offsets.p, 0, 4));
        *out = hold.release();
        return GM_OK;
    }
    gm::DevBuf pairs, palt, new_id, new_deg;
    int rc;
    if ((rc = pairs.alloc(n * 8)) || (rc = palt.alloc(n * 8)) || (rc = new_id.alloc(n * 4)) ||
        (rc = new_deg.alloc((n + 1) * 4)))
        return fail(rc);
    hipLaunchKernelGGL(degree_pairs_kernel, dim3(stream_grid(n)), dim3(256), 0, 0, g->offsets, (uint32_t)n,
                       pairs.as<uint64_t>());
    {
        rocprim::double_buffer<uint64_t> db(pairs.as<uint64_t>(), palt.as<uint64_t>());
        size_t tmp_bytes = 0;
        hipError_t e = rocprim::radix_sort_keys_desc(nullptr, tmp_bytes, db, n, 0u, 64u, (hipStream_t)0);
        gm::DevBuf tmp;
        if (e == hipSuccess && (rc = tmp.alloc(tmp_bytes)))
            return fail(rc);
        if (e == hipSuccess)
            e = rocprim::radix_sort_keys_desc(tmp.p, tmp_bytes, db, n, 0u, 64u, (hipStream_t)0);
        if (e == hipSuccess)
            e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            gm::set_error("gm_csr_relabel_by_degree: sort failed: %s", hipGetErrorString(e));
            return fail(GM_ERR_HIP);
        }
        if (db.current() != pairs.as<uint64_t>())
            std::swap(pairs, palt);
    }
    hipLaunchKernelGGL(unzip_pairs_kernel, dim3(stream_grid(n + 1)), dim3(256), 0, 0, pairs.as<uint64_t>(), (uint32_t)n,
                       new_id.as<uint32_t>(), new_deg.as<uint32_t>());
    if ((rc = exclusive_scan_u32(new_deg.as<uint32_t>(), c->own_offsets.as<uint32_t>(), n + 1)))
        return fail(rc);
    if (m) {
        gm::DevBuf keys, kalt;
        if ((rc = keys.alloc_big(m * 8)) || (rc = kalt.alloc_big(m * 8)))
            return fail(rc);
        hipLaunchKernelGGL(relabel_keys_kernel, dim3(stream_grid(m)), dim3(256), 0, 0, g->offsets, g->targets,
                           (uint32_t)n, m, new_id.as<uint32_t>(), keys.as<uint64_t>());
        if ((rc = radix_sort_keys_inplace<uint64_t>(keys, kalt, m, 0, 32 + ceil_log2(n))))
            return fail(rc);
        hipLaunchKernelGGL(targets_from_keys64_kernel, dim3(stream_grid(m)), dim3(256), 0, 0, keys.as<uint64_t>(), m,
                           c->own_targets.as<uint32_t>());
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            gm::set_error("gm_csr_relabel_by_degree: kernel failure");
            return fail(GM_ERR_HIP);
        }
    }
    if (new_id_out && hipMemcpy(new_id_out, new_id.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        gm::set_error("gm_csr_relabel_by_degree: read-back failed");
        return fail(GM_ERR_HIP);
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        gm::set_error("gm_csr_relabel_by_degree: sync failed");
        return fail(GM_ERR_HIP);
    }
    *out = hold.release();
    return GM_OK;
}
