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
#include <vector>

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

struct DeviceSwitch { // sets the current device for a scope (DeviceGuard below also reports failure)
    int prev = -1;
    explicit DeviceSwitch(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != dev)
            (void)hipSetDevice(dev);
        else
            prev = -1;
    }
    ~DeviceSwitch()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// arena.hip: the library's own supply of 64 MiB physical pieces for its large buffers
// Size of the arena's physical pieces (GM_ARENA_PIECE_MIB, read once; a power of two from 2 to 4096).
size_t arena_piece_bytes();
#define ARENA_PIECE (gm::arena_piece_bytes())
constexpr size_t ARENA_MIN = (size_t)128 << 20; // smaller buffers stay with hipMalloc
struct ArenaPiece {
    hipMemGenericAllocationHandle_t handle;
    uint64_t serial; // creation order
};
// HIP loads the code object of a translation unit when the first kernel of it is launched: ~10 ms each for the larger ones,
// which made every algorithm's FIRST call on a process pay for its own (SSSP: 9.7 of a first call's 10.4 ms of set-up were the
// launch of its first kernel, tools/runs/r06_call11.sh).  The handle constructors load all of them once, where a graph is being
// built or uploaded anyway (GM_WARM=0: as before).
void warm_code_objects();
void warm_pagerank();
void warm_pagerank_pb();
void warm_wcc();
void warm_sssp();
void warm_tc();
void warm_multi();
// an environment variable that only measurements ever set (A/B records: CHANGELOG.md): absent in the product library
#ifdef GM_MEASURE
inline const char *measure_env(const char *name) { return getenv(name); }
#else
inline const char *measure_env(const char *) { return nullptr; }
#endif
bool arena_enabled(); // GM_ARENA=0: every buffer from hipMalloc
inline int &arena_site() // which part of the library is allocating (1 CSR build, 2 plan temporaries, 4 plan streams, 8 value stream)
{
    static thread_local int site = 0;
    return site;
}
inline bool arena_site_enabled(int site) // GM_ARENA_SITES=<mask> (debugging): which of them may use the arena
{
    const char *v = getenv("GM_ARENA_SITES");
    const int s = site ? site : arena_site();
    return !(v && *v) || s == 0 || (atoi(v) & s);
}
// `count` pieces; spread_seed != 0: a stratified pseudo-random subset of a free list of >= count x spread_factor pieces
// (of the free pieces created as numbers [serial_lo, serial_hi), when that range is given: no growth then)
int arena_take(int dev, size_t count, uint64_t spread_seed, size_t spread_factor, std::vector<ArenaPiece> &out,
               uint64_t serial_lo = 0, uint64_t serial_hi = ~0ull);
int arena_grow(int dev, size_t count, uint64_t *first_serial_out);
// virtual addresses out of the arena's own reservation (never returned to the runtime: see arena.hip)
int arena_va_alloc(int dev, size_t span, void **out);
void arena_va_free(int dev, void *ptr, size_t span);
void arena_give(int dev, std::vector<ArenaPiece> &pieces);
void arena_trim(int dev, size_t keep_bytes);
void arena_stats(int dev, uint64_t *out4);
void arena_va_stats(int dev, uint64_t *out3);
size_t arena_class_pieces(size_t count); // size classes of the arena's buffers: see arena.hip
// serials of the oldest and (one past) the newest piece of the free list, and how many are free
void arena_free_range(int dev, uint64_t *lo, uint64_t *hi, size_t *count);

// RAII device allocation; movable, not copyable.  Two backings: hipMalloc / hipFree, or (alloc_vmm) a reserved
// virtual range mapped chunk by chunk from physical allocations of the HIP virtual-memory API — the caller decides
// the size of the physical pieces and the order in which they appear in the range (pagerank_pb.hip: the value
// stream's sweep time depends on the pages it lands on).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> vmm; // physical pieces of a mapped range (empty: hipMalloc backing)
    size_t vmm_chunk = 0, vmm_span = 0;                // bytes per piece, bytes reserved
    std::vector<ArenaPiece> arena;                     // pieces borrowed from the arena (alloc_big); returned on release
    int arena_dev = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept { take(o); }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) {
            release();
            take(o);
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void take(DevBuf &o)
    {
        p = o.p, bytes = o.bytes, vmm = std::move(o.vmm), vmm_chunk = o.vmm_chunk, vmm_span = o.vmm_span;
        arena = std::move(o.arena), arena_dev = o.arena_dev;
        o.p = nullptr, o.bytes = 0, o.vmm.clear(), o.vmm_chunk = o.vmm_span = 0, o.arena.clear();
    }
    void release()
    {
        if (!arena.empty()) {
            // hipFree waits for the device; so does this (a kernel may still be reading the buffer)
            DeviceSwitch sw(arena_dev);
            (void)hipDeviceSynchronize();
            if (p) { // piece by piece, as they were mapped (one hipMemUnmap over the whole range left the later pieces mapped:
                     // the freed addresses then reached hipMalloc with the arena's memory still behind them)
                for (size_t i = 0; i < arena.size(); ++i)
                    (void)hipMemUnmap(static_cast<char *>(p) + i * ARENA_PIECE, ARENA_PIECE);
                arena_va_free(arena_dev, p, vmm_span); // back to the arena's own address space, not to the runtime
            }
            arena_give(arena_dev, arena);
            vmm_chunk = vmm_span = 0;
        } else if (vmm_span) {
            (void)hipDeviceSynchronize(); // as hipFree would: a kernel may still be using the buffer
            if (p) {
                for (size_t i = 0; i < vmm.size(); ++i)
                    (void)hipMemUnmap(static_cast<char *>(p) + i * vmm_chunk, vmm_chunk);
                // the range itself is left reserved (measurement paths only): freed ranges corrupt later allocations (arena.hip)
            }
            for (hipMemGenericAllocationHandle_t h : vmm)
                (void)hipMemRelease(h);
            vmm.clear();
            vmm_chunk = vmm_span = 0;
        } else if (p) {
            (void)hipFree(p);
        }
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t nbytes)
    {
        release();
        if (nbytes == 0)
            nbytes = 16; // keep pointers non-null for empty graphs
        hipError_t me = hipMalloc(&p, nbytes);
        if (me == hipErrorOutOfMemory && arena_enabled()) {
            // the arena's idle pieces are the library's own reserve: hand them back and try once more before reporting
            // a device that "has no memory"
            (void)hipGetLastError();
            int dev_now = 0;
            if (hipGetDevice(&dev_now) == hipSuccess) {
                (void)hipDeviceSynchronize();
                arena_trim(dev_now, 0);
                me = hipMalloc(&p, nbytes);
            }
        }
        GM_HIP(me);
        bytes = nbytes;
        // GM_POISON="<min bytes>,<max bytes>" (debugging): allocations in that size range start out as 0xFF bytes (f32 NaN)
        // instead of whatever the driver hands out (usually zeros) — finds code that reads what it never wrote
        if (const char *v = getenv("GM_POISON")) {
            unsigned long long lo = 0, hi = ~0ull;
            (void)sscanf(v, "%llu,%llu", &lo, &hi);
            if (nbytes >= lo && nbytes <= hi) {
                GM_HIP(hipMemset(p, 0xFF, nbytes));
                GM_HIP(hipDeviceSynchronize());
            }
        }
        return GM_OK;
    }
    // A large buffer from the arena's 64 MiB pieces (arena.hip); spread_seed != 0: pieces sampled from all over the
    // arena's free list, which is grown to spread_factor times the request first.  Small requests, and every request
    // under GM_ARENA=0, take the hipMalloc path.  Contents are NOT zero (hipMalloc does not promise that either).
    // serial_lo / serial_hi: only pieces created as numbers [lo, hi) (arena_grow); split_serial != 0: every other piece
    // of the buffer from below that serial, the others from it on
    // a temporary of a build step: from the arena's idle pieces (any of them) from 32 MiB up — a hipMalloc / hipFree pair
    // per temporary is what stalls the NEXT hipMalloc for hundreds of milliseconds (the driver clears freed VRAM first)
    int alloc_scratch(size_t nbytes)
    {
        if (arena_enabled() && nbytes >= ((size_t)32 << 20) && arena_site_enabled(2) &&
            alloc_from_arena(nbytes, 0, 1, 0, ~0ull, 0) == GM_OK)
            return GM_OK;
        return alloc(nbytes);
    }
    // split_serial != 0: half of the pieces from serials [serial_lo, older_hi ? older_hi : split_serial), half from
    // [split_serial, serial_hi), interleaved
    int alloc_big(size_t nbytes, uint64_t spread_seed = 0, size_t spread_factor = 4, uint64_t serial_lo = 0,
                  uint64_t serial_hi = ~0ull, uint64_t split_serial = 0, int site = 0, uint64_t older_hi = 0)
    {
        if (!arena_enabled() || nbytes < ARENA_MIN || !arena_site_enabled(site))
            return alloc(nbytes);
        // the arena is a matter of speed, not of function: if the virtual-memory path fails (no pieces left, a runtime
        // that refuses the mapping), the buffer comes from hipMalloc like every small one
        if (alloc_from_arena(nbytes, spread_seed, spread_factor, serial_lo, serial_hi, split_serial, older_hi) == GM_OK)
            return GM_OK;
        if (const char *v = getenv("GM_LOG"))
            if (*v && *v != '0')
                fprintf(stderr, "[graph_mi355x] arena: %s; falling back to hipMalloc for %zu bytes\n", gm_last_error(), nbytes);
        return alloc(nbytes);
    }
    int alloc_from_arena(size_t nbytes, uint64_t spread_seed, size_t spread_factor, uint64_t serial_lo, uint64_t serial_hi,
                         uint64_t split_serial, uint64_t older_hi = 0)
    {
        release();
        int dev = 0;
        GM_HIP(hipGetDevice(&dev));
        // whole size classes: the released range and piece set fit the next buffer of about this size (arena.hip)
        const size_t count = arena_class_pieces((nbytes + ARENA_PIECE - 1) / ARENA_PIECE), span = count * ARENA_PIECE;
        if (split_serial && spread_seed) {
            std::vector<ArenaPiece> older, newer;
            GM_TRY(arena_take(dev, count / 2, spread_seed, 1, older, serial_lo, older_hi ? older_hi : split_serial));
            int rc2 = arena_take(dev, count - count / 2, spread_seed + 1, 1, newer, split_serial, serial_hi);
            if (rc2 != GM_OK) {
                arena_give(dev, older);
                return rc2;
            }
            for (size_t i = 0; i < newer.size(); ++i) {
                arena.push_back(newer[i]);
                if (i < older.size())
                    arena.push_back(older[i]);
            }
        } else {
            GM_TRY(arena_take(dev, count, spread_seed, spread_factor, arena, serial_lo, serial_hi));
        }
        arena_dev = dev;
        void *base = nullptr;
        {
            const int rc_va = arena_va_alloc(dev, span, &base);
            if (rc_va != GM_OK) {
                arena_give(dev, arena);
                return rc_va;
            }
        }
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < count && e == hipSuccess; ++i) {
            e = hipMemMap(static_cast<char *>(base) + i * ARENA_PIECE, ARENA_PIECE, 0, arena[i].handle, 0);
            if (e != hipSuccess)
                for (size_t j = 0; j < i; ++j)
                    (void)hipMemUnmap(static_cast<char *>(base) + j * ARENA_PIECE, ARENA_PIECE);
        }
        if (e == hipSuccess) {
            hipMemAccessDesc desc{};
            desc.location.type = hipMemLocationTypeDevice;
            desc.location.id = dev;
            desc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base, span, &desc, 1);
            if (e != hipSuccess)
                for (size_t j = 0; j < count; ++j)
                    (void)hipMemUnmap(static_cast<char *>(base) + j * ARENA_PIECE, ARENA_PIECE);
        }
        if (e != hipSuccess) {
            gm::set_error("alloc_big(%zu bytes): %s", nbytes, hipGetErrorString(e));
            arena_va_free(dev, base, span);
            arena_give(dev, arena);
            return e == hipErrorOutOfMemory ? GM_ERR_NOMEM : GM_ERR_HIP;
        }
        p = base, bytes = span, vmm_span = span, vmm_chunk = ARENA_PIECE;
        return GM_OK;
    }
    // `nbytes` as pieces of `chunk` bytes (rounded up to the allocation granularity; 0: one piece), the range
    // aligned to `va_align` bytes (0: the granularity), pieces mapped in the order order[i] (null: ascending)
    int alloc_vmm(size_t nbytes, size_t chunk, size_t va_align, const uint32_t *order = nullptr)
    {
        release();
        int dev = 0;
        GM_HIP(hipGetDevice(&dev));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        GM_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0)
            gran = 2u << 20;
        if (nbytes == 0)
            nbytes = 16;
        if (chunk == 0 || chunk > nbytes)
            chunk = nbytes;
        chunk = (chunk + gran - 1) / gran * gran;
        const size_t count = (nbytes + chunk - 1) / chunk;
        const size_t span = count * chunk;
        if (va_align < gran)
            va_align = gran;
        void *base = nullptr;
        GM_HIP(hipMemAddressReserve(&base, span, va_align, nullptr, 0));
        p = base, vmm_span = span, vmm_chunk = chunk; // from here on release() undoes what has been done
        vmm.reserve(count);
        for (size_t i = 0; i < count; ++i) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
            if (e != hipSuccess) {
                gm::set_error("hipMemCreate(%zu bytes, piece %zu of %zu) failed: %s", chunk, i, count, hipGetErrorString(e));
                (void)hipMemAddressFree(base, span);
                p = nullptr, vmm_span = 0;
                release_handles();
                return e == hipErrorOutOfMemory ? GM_ERR_NOMEM : GM_ERR_HIP;
            }
            vmm.push_back(h);
        }
        for (size_t i = 0; i < count; ++i) {
            hipError_t e = hipMemMap(static_cast<char *>(base) + i * chunk, chunk, 0, vmm[order ? order[i] : i], 0);
            if (e != hipSuccess) {
                gm::set_error("hipMemMap failed: %s", hipGetErrorString(e));
                if (i)
                    (void)hipMemUnmap(base, i * chunk);
                (void)hipMemAddressFree(base, span);
                p = nullptr, vmm_span = 0;
                release_handles();
                return GM_ERR_HIP;
            }
        }
        hipMemAccessDesc desc{};
        desc.location = prop.location;
        desc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e = hipMemSetAccess(base, span, &desc, 1);
        if (e != hipSuccess) {
            gm::set_error("hipMemSetAccess failed: %s", hipGetErrorString(e));
            release();
            return GM_ERR_HIP;
        }
        bytes = span;
        return GM_OK;
    }
    // `pool` physical pieces of `chunk` bytes are created one after the other — consecutive allocations of a device
    // are mostly neighbours in physical memory — and the range is mapped from the pieces pick[0], pick[1], ... of that
    // sequence; the others are released again.  What it is for: DESIGN 4.1 ("where the pages are").
    int alloc_vmm_pool(size_t nbytes, size_t chunk, size_t pool, const std::vector<size_t> &pick)
    {
        release();
        int dev = 0;
        GM_HIP(hipGetDevice(&dev));
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        const size_t count = pick.size(), span = count * chunk;
        std::vector<hipMemGenericAllocationHandle_t> all;
        all.reserve(pool);
        hipError_t e = hipSuccess;
        for (size_t i = 0; i < pool && e == hipSuccess; ++i) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, chunk, &prop, 0);
            if (e == hipSuccess)
                all.push_back(h);
        }
        void *base = nullptr;
        if (e == hipSuccess)
            e = hipMemAddressReserve(&base, span, chunk < ((size_t)1 << 30) ? chunk : ((size_t)1 << 30), nullptr, 0);
        if (e != hipSuccess) {
            gm::set_error("alloc_vmm_pool: %s (%zu pieces of %zu bytes for %zu bytes)", hipGetErrorString(e), pool, chunk, nbytes);
            for (hipMemGenericAllocationHandle_t h : all)
                (void)hipMemRelease(h);
            return e == hipErrorOutOfMemory ? GM_ERR_NOMEM : GM_ERR_HIP;
        }
        p = base, vmm_span = span, vmm_chunk = chunk;
        std::vector<bool> used(all.size(), false);
        for (size_t i = 0; i < count; ++i) {
            used[pick[i]] = true;
            vmm.push_back(all[pick[i]]);
        }
        for (size_t k = 0; k < all.size(); ++k)
            if (!used[k])
                (void)hipMemRelease(all[k]);
        for (size_t i = 0; i < count && e == hipSuccess; ++i)
            e = hipMemMap(static_cast<char *>(base) + i * chunk, chunk, 0, vmm[i], 0);
        hipMemAccessDesc desc{};
        desc.location = prop.location;
        desc.flags = hipMemAccessFlagsProtReadWrite;
        if (e == hipSuccess)
            e = hipMemSetAccess(base, span, &desc, 1);
        if (e != hipSuccess) {
            gm::set_error("alloc_vmm_pool: map failed: %s", hipGetErrorString(e));
            release();
            return GM_ERR_HIP;
        }
        bytes = span;
        return GM_OK;
    }
    static size_t vmm_granularity()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0)
            gran = 2u << 20;
        return gran;
    }
    // pieces first, first + stride, first + 2 stride, ... (measurement)
    int alloc_vmm_strided(size_t nbytes, size_t chunk, size_t pool, size_t first, size_t stride)
    {
        const size_t gran = vmm_granularity();
        chunk = (chunk + gran - 1) / gran * gran;
        const size_t count = (nbytes + chunk - 1) / chunk;
        if (stride == 0)
            stride = 1;
        if (pool < first + (count - 1) * stride + 1)
            pool = first + (count - 1) * stride + 1;
        std::vector<size_t> pick(count);
        for (size_t i = 0; i < count; ++i)
            pick[i] = first + i * stride;
        return alloc_vmm_pool(nbytes, chunk, pool, pick);
    }
    // a pseudo-random subset of `factor` times as many pieces as the range needs
    int alloc_spread(size_t nbytes, size_t chunk, size_t factor, uint64_t seed)
    {
        const size_t gran = vmm_granularity();
        chunk = (chunk + gran - 1) / gran * gran;
        if (nbytes == 0)
            nbytes = 16;
        const size_t count = (nbytes + chunk - 1) / chunk;
        const size_t pool = count * (factor ? factor : 1);
        std::vector<size_t> idx(pool);
        for (size_t i = 0; i < pool; ++i)
            idx[i] = i;
        uint64_t state = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        for (size_t i = 0; i < count; ++i) { // partial Fisher-Yates: the first `count` entries are the subset
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            std::swap(idx[i], idx[i + z % (pool - i)]);
        }
        idx.resize(count);
        return alloc_vmm_pool(nbytes, chunk, pool, idx);
    }
    void release_handles()
    {
        for (hipMemGenericAllocationHandle_t h : vmm)
            (void)hipMemRelease(h);
        vmm.clear();
        vmm_chunk = 0;
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
inline bool log_enabled() // read at every phase boundary (a getenv), so that a host can switch it on for one call
{
    const char *v = getenv("GM_LOG");
    return v && *v && *v != '0';
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
struct MultiState;  // what gm_page_rank_multi derives from a graph and a device list (multi.hip), parked likewise
struct MultiStateDeleter {
    void operator()(MultiState *p) const; // multi.hip (the type is complete there)
};
struct WccScratch {
    DevBuf work;   // chunk count + chunk items + sample buffer (wcc.hip:wcc_device)
    DevBuf labels; // u32[n] of gm_wcc_afforest / gm_wcc_baseline
};
// every adjacency list of a weighted CSR once more, ordered by weight (sssp.hip: the edges a node relaxes while its phase is
// busy are then a PREFIX of its list, the others a suffix); immutable once built, kept in the handle like the PageRank plan
struct SsspOrder {
    DevBuf targets, weights;
    // ... and transposed (in_off u32[n + 1], in_edge uint2[m] = (source, weight bits) of every node's in-edges, in no
    // particular order): the far round pulls (sssp.hip)
    DevBuf in_off, in_edge;
};
struct SsspScratch {
    DevBuf dist, flags, wmin, hflags, fflags, settled, done, ctrl, chunks, queues, queues_init;
    uint64_t caps_key = 0; // (chunk_edges << 32 | coop) + 1 the sub-queue capacities in queues_init were computed for; 0: none
    PinnedBuf hctrl;
    size_t items = 0; // capacity of `chunks` in work items; 0: not (completely) allocated
};
} // namespace gm

// The opaque handle of include/graph_mi355x.h.
namespace gm {
// GM_PB_HUB_LEAVES: a row with at least this many sources that have at most one in-edge themselves is summed in the reference's order whatever
// its length (pagerank_pb.hip: pb_leafflag_kernel); 0 = the rule is off.  One reader for the plan builder and the partitioned front.
constexpr uint32_t kHubLeavesDefault = 512;
inline uint32_t hub_leaves_threshold()
{
    const char *v = getenv("GM_PB_HUB_LEAVES");
    return v && *v ? (uint32_t)atoi(v) : kHubLeavesDefault;
}
} // namespace gm

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
    mutable std::shared_ptr<const gm::SsspOrder> sssp_order; // likewise; built by the SECOND gm_sssp_delta_stepping call
    mutable std::atomic<uint64_t> sssp_calls{0};             // gm_sssp_delta_stepping calls seen by this handle
    mutable std::mutex sssp_build_mu;                        // one builder of sssp_order per handle (a second caller waits and finds it)
    mutable std::atomic<int> sssp_order_failed{0};           // 1: the build found no room; not retried until gm_csr_trim
    mutable std::unique_ptr<gm::WccScratch> wcc_scratch;   // likewise
    mutable std::shared_ptr<gm::PrCallState> pr_call;      // likewise (stream, vectors, engine + its scratch)
    mutable std::shared_ptr<const gm::TcDag> tc_dag;       // the DAG of lower prefixes + list records of gm_triangle_count
    mutable std::unique_ptr<gm::MultiState, gm::MultiStateDeleter> multi; // gm_page_rank_multi's resident run (in-CSR handle)
    // (threshold << 1 | answer) of the last look: does some row have >= GM_PB_HUB_DEG entries?  -1: not looked at yet
    mutable std::atomic<long long> long_rows{-1};
    // gm_csr_set_source_flags: one byte per entry of the x vector the rows' lists index (a partition slice: exchange slots), non-zero =
    // "this source has at most one in-edge" — what a slice cannot see in its own offsets (the plan builder's rule for rows of constant terms)
    gm::DevBuf source_flags;
    uint64_t source_flags_len = 0;
};
