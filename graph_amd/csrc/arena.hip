// arena.hip — the library's own supply of physical device memory for its LARGE buffers (gfx950 / MI355X only).
//
// Why (measured in round 3, profiles/r03_placement_*.txt):
//  * Where a buffer's pages lie decides up to 35 % of a streaming kernel's time.  The value stream of the PageRank
//    sweep (3.6 GB at RMAT scale 26, written in 8 M scattered runs) takes 1.66 ms to write when it sits in ONE
//    stretch of physical memory — what a fresh hipMalloc returns — 1.31 ms when it straddles two, and 1.23-1.30 ms
//    when it is mapped from 64 MiB pieces drawn from all over a few dozen GiB: the DRAM banks a stretch of physical
//    memory can use are chosen by high address bits, and scattered row activations need all of them.
//  * hipMalloc after large hipFree calls stalls for seconds (the driver hands freed VRAM back only after clearing it):
//    memory that never goes back to the driver is never waited for.
// So large buffers are mapped (HIP virtual-memory API) from 64 MiB physical pieces that the arena creates on demand
// and keeps when a buffer is released (up to GM_ARENA_KEEP_GIB, default 32, and never more than a quarter of what the
// device has free or idle; gm_trim() releases them).  Sizes come in CLASSES (arena_class_pieces: three mantissa bits, at
// most 12.5 % above the request), so that the address ranges and piece sets of released buffers fit the next request of
// about the same size: a long-lived process with graphs of many sizes reuses a bounded set of ranges.  A buffer asks
// either for any pieces or for a SPREAD subset: one pseudo-random piece out of every stratum of the free list in
// creation order — consecutive creations are mostly neighbours in physical memory, so the subset samples the whole
// stretch the arena has seen.
#include "common.hpp"

#include <algorithm>

namespace gm {

namespace {

// Virtual addresses are the arena's own too: ONE large reservation per device (more when it runs out), carved up here and
// never handed back.  Measured on ROCm 7.0: after hipMemAddressFree of a range that had held mappings, later hipMalloc
// allocations of the process turned up with garbage in them (PageRank returned NaN at RMAT scale 21-24; with the freed
// ranges leaked instead, every result was right) — the runtime reuses the addresses while something still remembers
// the old mappings.  Address space is not scarce (the default reservation is 1 TiB of a 2^47-byte space).
struct VaRange {
    char *base = nullptr;
    size_t size = 0, bump = 0;
};

struct Arena {
    std::mutex mu;
    uint64_t va_reserved = 0;                  // bytes of address space reserved so far (never returned: see above)
    std::vector<VaRange> va;                   // reservations, the last one is the one being carved up
    std::multimap<size_t, char *> va_free;     // returned stretches by size (buffers of a few recurring sizes come and go)
    std::vector<ArenaPiece> free_list; // ascending serial = creation order
    uint64_t next_serial = 0;
    uint64_t alive = 0; // pieces created and not released to the driver
    uint64_t created = 0, reused = 0;
};

Arena &arena_of(int dev)
{
    // never destroyed: a handle released during static destruction (a global graph object of the host program) still
    // finds its arena
    static std::mutex &mu = *new std::mutex;
    static std::map<int, std::unique_ptr<Arena>> &arenas = *new std::map<int, std::unique_ptr<Arena>>;
    std::lock_guard<std::mutex> lock(mu);
    std::unique_ptr<Arena> &a = arenas[dev];
    if (!a)
        a.reset(new Arena());
    return *a;
}

uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int create_pieces(Arena &a, int dev, size_t count)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    for (size_t i = 0; i < count; ++i) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, ARENA_PIECE, &prop, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("arena: hipMemCreate(64 MiB) failed after %llu pieces: %s", (unsigned long long)a.alive, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? GM_ERR_NOMEM : GM_ERR_HIP;
        }
        a.free_list.push_back(ArenaPiece{h, a.next_serial++});
        ++a.alive;
        ++a.created;
    }
    return GM_OK;
}

} // namespace

int arena_va_alloc(int dev, size_t span, void **out)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    // exact size only: a stretch is reused for a buffer of the size it held before.  (Best fit with split remainders was
    // tried — tools/runs/r03_call60.sh — and a process then hung: mapping into a PART of a range that had carried a larger
    // mapping is one more thing this runtime does not take well.  Buffers come in a few recurring sizes.)
    auto it = a.va_free.find(span);
    if (it != a.va_free.end()) {
        *out = it->second;
        a.va_free.erase(it);
        return GM_OK;
    }
    if (a.va.empty() || a.va.back().size - a.va.back().bump < span) {
        const char *v = getenv("GM_ARENA_VA_GIB");
        size_t want = (size_t)(v && atol(v) > 0 ? atol(v) : 1024) << 30;
        while (want < 2 * span)
            want *= 2;
        void *base = nullptr;
        hipError_t e = hipErrorOutOfMemory;
        for (; want >= span; want /= 2) { // a smaller reservation if the large one is refused
            e = hipMemAddressReserve(&base, want, ARENA_PIECE, nullptr, 0);
            if (e == hipSuccess)
                break;
            (void)hipGetLastError();
        }
        if (e != hipSuccess) {
            set_error("arena: hipMemAddressReserve(%zu bytes) failed: %s", span, hipGetErrorString(e));
            return GM_ERR_NOMEM;
        }
        if (!a.va.empty() && a.va.back().size > a.va.back().bump) // what is left of the old reservation stays usable
            a.va_free.emplace(a.va.back().size - a.va.back().bump, a.va.back().base + a.va.back().bump);
        a.va.push_back(VaRange{static_cast<char *>(base), want, 0});
        a.va_reserved += want;
    }
    VaRange &r = a.va.back();
    *out = r.base + r.bump;
    r.bump += span;
    return GM_OK;
}

void arena_va_free(int dev, void *ptr, size_t span)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    a.va_free.emplace(span, static_cast<char *>(ptr));
}

size_t arena_piece_bytes()
{
    static const size_t bytes = [] {
        const char *v = getenv("GM_ARENA_PIECE_MIB");
        long mib = v && *v ? atol(v) : 64;
        if (mib < 2 || mib > 4096 || (mib & (mib - 1)))
            mib = 64;
        return (size_t)mib << 20;
    }();
    return bytes;
}

bool arena_enabled()
{
    static const bool on = [] {
        const char *v = getenv("GM_ARENA");
        return !(v && *v == '0');
    }();
    return on;
}

// idle pieces the arena may hold on to: GM_ARENA_KEEP_GIB (default 32), but at most a quarter of the memory that is free or
// idle in the arena right now — on a device other tenants have filled (torch's caching allocator, other graphs) the reserve
// shrinks with what is left.  `idle_bytes`: the arena's own idle pieces (they do not show up as free memory).
size_t arena_keep_bytes(size_t idle_bytes)
{
    const char *v = getenv("GM_ARENA_KEEP_GIB");
    const long gib = v && *v ? atol(v) : 32;
    size_t keep = (size_t)(gib < 0 ? 0 : gib) << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t quarter = (free_b + idle_bytes) / 4;
        keep = keep < quarter ? keep : quarter;
    } else {
        (void)hipGetLastError();
    }
    return keep;
}

// pieces of the size class of a request of `count` pieces: up to 16 exact, beyond that three mantissa bits (the next multiple
// of 2^(floor(log2 count) - 3): at most 12.5 % more)
size_t arena_class_pieces(size_t count)
{
    if (count <= 16)
        return count;
    int top = 0;
    while ((count >> (top + 1)) != 0)
        ++top;
    const size_t step = (size_t)1 << (top - 3);
    return (count + step - 1) / step * step;
}

int arena_take(int dev, size_t count, uint64_t spread_seed, size_t spread_factor, std::vector<ArenaPiece> &out,
               uint64_t serial_lo, uint64_t serial_hi)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    out.clear();
    const bool ranged = serial_lo != 0 || serial_hi != ~0ull;
    size_t want_free = spread_seed ? count * (spread_factor ? spread_factor : 1) : count;
    if (!ranged && a.free_list.size() < want_free) {
        // The pool beyond the request itself (a spread draw wants `spread_factor` times the pieces to choose from) is a
        // matter of speed: it never takes more than half of what the device has free, and if the device runs out on the
        // way the surplus goes back at once — the arena must not be what makes a later hipMalloc fail.
        size_t free_b = 0, total_b = 0;
        if (want_free > count && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t have = a.free_list.size(), need = count > have ? count - have : 0;
            const size_t extra_cap = free_b / 2 / ARENA_PIECE > need ? free_b / 2 / ARENA_PIECE - need : 0;
            const size_t capped = (have > count ? have : count) + extra_cap;
            want_free = want_free < capped ? want_free : capped;
        }
        const int rc = a.free_list.size() < want_free ? create_pieces(a, dev, want_free - a.free_list.size()) : GM_OK;
        if (rc != GM_OK) {
            (void)hipGetLastError();
            if (a.free_list.size() < count)
                return rc;
            // a smaller pool than asked for still serves the request; what was created beyond it is handed back so that
            // at least 1/16 of the device stays free
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                while (a.free_list.size() > count && free_b < total_b / 16) {
                    (void)hipMemRelease(a.free_list.back().handle);
                    a.free_list.pop_back();
                    --a.alive;
                    free_b += ARENA_PIECE;
                }
        }
    }
    if (a.free_list.size() < count) {
        set_error("arena: %zu free pieces, %zu wanted", a.free_list.size(), count);
        return GM_ERR_NOMEM;
    }
    if (!spread_seed) { // any pieces: the most recently returned ones
        a.reused += count;
        out.assign(a.free_list.end() - (ptrdiff_t)count, a.free_list.end());
        a.free_list.resize(a.free_list.size() - count);
        return GM_OK;
    }
    // one pseudo-random piece from every stratum of the free list (kept in creation order) — of its part with serials
    // in [serial_lo, serial_hi) when a range is given
    std::sort(a.free_list.begin(), a.free_list.end(), [](const ArenaPiece &x, const ArenaPiece &y) { return x.serial < y.serial; });
    size_t f0 = 0, f1 = a.free_list.size();
    while (f0 < f1 && a.free_list[f0].serial < serial_lo)
        ++f0;
    while (f1 > f0 && a.free_list[f1 - 1].serial >= serial_hi)
        --f1;
    if (f1 - f0 < count) {
        set_error("arena: %zu free pieces in the serial range, %zu wanted", f1 - f0, count);
        return GM_ERR_NOMEM;
    }
    a.reused += count;
    const size_t F = a.free_list.size(), W = f1 - f0;
    std::vector<bool> taken(F, false);
    for (size_t i = 0; i < count; ++i) {
        const size_t lo = f0 + i * W / count, hi = f0 + (i + 1) * W / count; // hi > lo: W >= count
        const size_t k = lo + (size_t)(mix64(spread_seed * 0x9E3779B97F4A7C15ull + i) % (hi - lo));
        taken[k] = true;
        out.push_back(a.free_list[k]);
    }
    // the stream's consecutive pieces should not be physical neighbours either: shuffle the order they are mapped in
    for (size_t i = count; i > 1; --i)
        std::swap(out[i - 1], out[mix64(spread_seed ^ (0xD1B54A32D192ED03ull * i)) % i]);
    size_t w = 0;
    for (size_t k = 0; k < F; ++k)
        if (!taken[k])
            a.free_list[w++] = a.free_list[k];
    a.free_list.resize(w);
    return GM_OK;
}

static void trim_locked(Arena &a, size_t keep_bytes)
{
    const size_t keep = keep_bytes / ARENA_PIECE;
    while (a.free_list.size() > keep) { // the oldest returns go first
        (void)hipMemRelease(a.free_list.front().handle);
        a.free_list.erase(a.free_list.begin());
        --a.alive;
    }
}

void arena_give(int dev, std::vector<ArenaPiece> &pieces)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    a.free_list.insert(a.free_list.end(), pieces.begin(), pieces.end());
    pieces.clear();
    const size_t idle = a.free_list.size() * ARENA_PIECE, keep = arena_keep_bytes(idle);
    if (idle > keep + (keep >> 2)) // hysteresis: trim in batches
        trim_locked(a, keep);
}

void arena_trim(int dev, size_t keep_bytes)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    trim_locked(a, keep_bytes);
}

// `count` more pieces in the free list (newer serials than any before); *first_serial_out = the first of them
int arena_grow(int dev, size_t count, uint64_t *first_serial_out)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    if (first_serial_out)
        *first_serial_out = a.next_serial;
    // an experiment's memory (pb_scratch_create tries fresh stretches when every candidate of the pool is slow): never more
    // than half of what the device has free right now — the plain allocations that follow must still fit
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return GM_ERR_NOMEM;
    }
    if (count * ARENA_PIECE > free_b / 2)
        return GM_ERR_NOMEM;
    return create_pieces(a, dev, count);
}

void arena_stats(int dev, uint64_t *out4)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    out4[0] = a.alive * ARENA_PIECE;
    out4[1] = a.free_list.size() * ARENA_PIECE;
    out4[2] = a.created;
    out4[3] = a.reused;
}

void arena_va_stats(int dev, uint64_t *out3)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    uint64_t idle = 0;
    for (const auto &kv : a.va_free)
        idle += kv.first;
    uint64_t unused = 0; // never handed out: the rest of the reservation being carved up
    if (!a.va.empty())
        unused = a.va.back().size - a.va.back().bump;
    out3[0] = a.va_reserved;
    out3[1] = idle;
    out3[2] = unused;
}

void arena_free_range(int dev, uint64_t *lo, uint64_t *hi, size_t *count)
{
    Arena &a = arena_of(dev);
    std::lock_guard<std::mutex> lock(a.mu);
    *lo = ~0ull, *hi = 0, *count = a.free_list.size();
    for (const ArenaPiece &p : a.free_list) {
        *lo = p.serial < *lo ? p.serial : *lo;
        *hi = p.serial + 1 > *hi ? p.serial + 1 : *hi;
    }
    if (a.free_list.empty())
        *lo = 0;
}

} // namespace gm

// Releases what the library holds in reserve on `device` (-1: the current device): the arena's free pieces.  Buffers in
// use (graphs, plans, parked scratch: see gm_csr_trim) are not touched.
GM_API int gm_trim(int device)
{
    int dev = device;
    if (dev < 0)
        GM_HIP(hipGetDevice(&dev));
    gm::DeviceGuard guard(dev);
    GM_HIP(hipDeviceSynchronize());
    gm::arena_trim(dev, 0);
    return GM_OK;
}

// bytes_out[4]: bytes of device memory the arena holds, bytes of it not in use, pieces created, pieces handed out
GM_API int gm_arena_info(int device, uint64_t *bytes_out)
{
    GM_CHECK(bytes_out, GM_ERR_INVALID, "gm_arena_info: null argument");
    int dev = device;
    if (dev < 0)
        GM_HIP(hipGetDevice(&dev));
    gm::arena_stats(dev, bytes_out);
    return GM_OK;
}

// bytes_out[3]: bytes of virtual address space the arena has reserved (never returned to the runtime), bytes of it in
// released ranges waiting for a request of their size class, bytes never handed out yet
GM_API int gm_arena_va_info(int device, uint64_t *bytes_out)
{
    GM_CHECK(bytes_out, GM_ERR_INVALID, "gm_arena_va_info: null argument");
    int dev = device;
    if (dev < 0)
        GM_HIP(hipGetDevice(&dev));
    gm::arena_va_stats(dev, bytes_out);
    return GM_OK;
}
