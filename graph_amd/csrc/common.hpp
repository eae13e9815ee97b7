// common.hpp — shared host-side plumbing of libgraph_mi355x (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>

#include "../../include/graph_mi355x.h"

#define GM_API extern "C" __attribute__((visibility("default")))

namespace gm {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

struct HipFail {
    hipError_t err;
};

#define GM_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t gm_e_ = (expr);                                                                \
        if (gm_e_ != hipSuccess) {                                                                \
            gm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(gm_e_), __FILE__,     \
                          __LINE__);                                                              \
            return gm_e_ == hipErrorOutOfMemory ? GM_ERR_NOMEM : GM_ERR_HIP;                       \
        }                                                                                         \
    } while (0)

#define GM_CHECK(cond, status, ...)                                                               \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            gm::set_error(__VA_ARGS__);                                                           \
            return (status);                                                                      \
        }                                                                                         \
    } while (0)

#define GM_TRY(expr)                                                                              \
    do {                                                                                          \
        int gm_s_ = (expr);                                                                       \
        if (gm_s_ != GM_OK)                                                                       \
            return gm_s_;                                                                         \
    } while (0)

// RAII device allocation (hipMalloc / hipFree); movable, not copyable.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t nbytes)
    {
        release();
        if (nbytes == 0)
            nbytes = 16; // keep pointers non-null for empty graphs
        GM_HIP(hipMalloc(&p, nbytes));
        bytes = nbytes;
        return GM_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Pinned host scratch for small read-backs (error values, counters).
struct PinnedBuf {
    void *p = nullptr;
    ~PinnedBuf()
    {
        if (p)
            (void)hipHostFree(p);
    }
    int alloc(size_t nbytes)
    {
        GM_HIP(hipHostMalloc(&p, nbytes, hipHostMallocDefault));
        return GM_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess)
            ok = true;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

inline unsigned div_up(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// GM_LOG=1: phase timings on stderr, the counterpart of the reference's `log::info!` lines
// (page_rank.rs:95-100, wcc.rs:132-182, sssp.rs:99).  Costs a stream synchronisation per phase.
inline bool log_enabled()
{
    static const bool on = [] {
        const char *v = getenv("GM_LOG");
        return v && *v && *v != '0';
    }();
    return on;
}

struct PhaseTimer { // wall clock around stream-ordered work; only active under GM_LOG
    hipStream_t st;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(hipStream_t s) : st(s)
    {
        if (log_enabled()) {
            (void)hipStreamSynchronize(st);
            t0 = std::chrono::steady_clock::now();
        }
    }
    void done(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        if (!log_enabled())
            return;
        (void)hipStreamSynchronize(st);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        fprintf(stderr, "[graph_mi355x] %s took %.3f ms\n", buf, ms);
        t0 = std::chrono::steady_clock::now();
    }
};

} // namespace gm

namespace gm {
struct PbPlan; // propagation-blocking layout of a CSR (pagerank_pb.hip); immutable once built

// Working buffers of one gm_sssp_delta_stepping call (sssp.hip).  A call takes the set parked in the CSR handle (or
// allocates one) and parks it again when it returns: hipMalloc / hipFree of ~200 MB cost more than a millisecond
// each and a graph is usually queried from many start nodes.
struct TcDag;       // what gm_triangle_count derives from the graph alone (tc.hip); immutable once built
struct PrCallState; // what one gm_page_rank call allocates (pagerank.hip), parked in the handle between calls
struct WccScratch {
    DevBuf work;   // chunk count + chunk items + sample buffer (wcc.hip:wcc_device)
    DevBuf labels; // u32[n] of gm_wcc_afforest / gm_wcc_baseline
};
struct SsspScratch {
    DevBuf dist, flags, wmin, hflags, settled, ctrl, chunks, queues;
    PinnedBuf hctrl;
    size_t items = 0; // capacity of `chunks` in work items; 0: not (completely) allocated
};
} // namespace gm

// The opaque handle of include/graph_mi355x.h.
struct gm_csr {
    uint64_t n = 0;
    uint64_t m = 0;
    int device = 0;
    const uint32_t *offsets = nullptr; // n+1
    const uint32_t *targets = nullptr; // m
    const float *weights = nullptr;    // m or null
    bool owns = false;
    gm::DevBuf own_offsets, own_targets, own_weights;
    // Derived, immutable layouts built on first use and kept for the lifetime of the handle ("upload
    // once"): PageRank's propagation-blocking plan, keyed by the length of the x vector it was built for.
    mutable std::mutex cache_mu;
    mutable std::map<uint64_t, std::shared_ptr<const gm::PbPlan>> pb_plans;
    mutable std::atomic<uint64_t> page_rank_calls{0}; // gm_page_rank calls seen by this handle (engine choice); calls may run concurrently
    mutable std::atomic<int> weights_ok{0};                // 1: gm_sssp_delta_stepping has seen that no weight is negative or NaN
    mutable std::unique_ptr<gm::SsspScratch> sssp_scratch; // parked between calls (under cache_mu)
    mutable std::unique_ptr<gm::WccScratch> wcc_scratch;   // likewise
    mutable std::shared_ptr<gm::PrCallState> pr_call;      // likewise (stream, vectors, engine + its scratch)
    mutable std::shared_ptr<const gm::TcDag> tc_dag;       // the DAG of lower prefixes + list records of gm_triangle_count
    mutable std::atomic<int> long_rows{-1};           // 1: some row has >= GM_PB_HUB_DEG entries (-1: not looked at yet)
};
