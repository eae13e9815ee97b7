// sssp.hip — delta-stepping single-source shortest paths on a device-resident weighted out-CSR.
//
// Replaces crates/algos/src/sssp.rs:38-204.  The reference's intended result is the least fixed point of
// d[v] = min(d[u] (+) w(u,v)) under f32 round-to-nearest addition, which is monotone — so the
// distances do not depend on the relaxation schedule and this kernel returns exactly that fixed point,
// bit for bit, although its buckets are organised differently.  (The reference itself misses the fixed
// point on inputs where its stale-entry test `d >= delta * bin` (:126) disagrees in f32 with the bin it
// chose as (d/delta) as usize (:192), e.g. d = 13.5, delta = 0.3: the node is skipped and its edges are
// never relaxed.  There is no such test here)
//   * distances are kept as u32 bit patterns (non-negative f32 order == unsigned order) and
//     relaxed with atomicMin — the reference's CAS-min loop (sssp.rs:180-202) in one instruction;
//   * instead of per-thread bins copied into a shared frontier (sssp.rs:85-94) there is one bit per
//     node: set when the node's distance improved since its edges were last relaxed.  A round clears
//     and relaxes every flagged node whose distance is <= the current threshold; the others simply
//     stay flagged (nothing is rewritten for them).  No fences: "distance improved, then flag set" and
//     "flag cleared, then distance read" are each ordered by the returned value of the first atomic, so
//     an improvement is never lost.  When a phase has run dry the threshold advances to (minimum pending
//     distance + width) — the role of the reference's min_non_empty_bin (sssp.rs:159-168); `delta` only
//     seeds the schedule (first step delta/32, then adapted to the work of each phase, see
//     gm_sssp_delta_stepping) and never changes the result;
//   * the bucket bookkeeping lives on the device (sssp_finish_kernel): the host enqueues rounds in
//     batches and reads one flag per batch instead of synchronising after every round;
//   * INF = f32::MAX (sssp.rs:12), never +inf.
// Kernels of a round: sssp_round_kernel (a wavefront per 1024 nodes: opens the flag words that can hold a near node,
// relaxes lists of <= 32 edges flattened over its lanes, queues work items for the longer ones), sssp_chunk_kernel
// (one wavefront per 256-edge item, whole grid; light / heavy split), sssp_finish_kernel (the pending minimum after a
// phase's heavy round, then the threshold bookkeeping).
#include "common.hpp"
#include "device_utils.hpp"

#include <rocprim/rocprim.hpp>

#include <cstdlib>

namespace {

using namespace gm;

constexpr int SSSP_BLOCK = 256;
constexpr uint32_t SSSP_COOP = 32; // lists longer than this are relaxed by the whole wavefront
constexpr uint32_t SSSP_INF_BITS = 0x7F7FFFFFu; // f32::MAX
constexpr uint32_t NO_BUCKET = 0xFFFFFFFFu;

struct RelaxOut {
    uint32_t again; // some distance at or below the threshold improved: the phase has not run dry
};

// thr: bit pattern of the current distance threshold (non-negative f32 order == unsigned order).
// `pre` is dist[t] as read by the caller's batched pre-check (several independent random reads in
// flight per lane); the atomic only runs when the candidate still looks like an improvement.
__device__ __forceinline__ void relax_checked(uint32_t *dist, uint32_t *flags, uint32_t *wmin, uint32_t nb, uint32_t pre,
                                              uint32_t t, uint32_t thr, RelaxOut &ro)
{
    if (nb < pre) {
        const uint32_t old = atomicMin(&dist[t], nb);
        if (nb < old) { // issued only once the atomicMin has returned: distance, then flag, then word summary
            uint32_t zero = atomicOr(&flags[t >> 5], 1u << (t & 31u));
            asm volatile("v_and_b32 %0, 0, %0" : "+v"(zero)); // 0, but only known once the atomicOr has returned
            atomicMin(&wmin[(t >> 5) + zero], nb);
            if (nb <= thr)
                ro.again = 1;
        }
    }
}

// relaxes edges i = first, first + step, ... < end of one source at distance du, SSSP_MLP at a time.
// `which` (uniform): R_LIGHT = only the edges whose candidate lands at or below the threshold (they can make a node
// of the running phase move again), R_HEAVY = only the others (relaxed once per phase, when it has run dry, with the
// source's final distance), R_ALL = every edge.  The targets and weights of the skipped edges are streamed, their
// distances not probed.
// R_NEAR (lists ordered by weight, a cut above the threshold): the heavy edges up to the cut, thr < candidate <= cut;
// with R_HEAVY and cut != NO_BUCKET... the far round: only the candidates beyond `cut`.
enum : int { R_ALL = 0, R_LIGHT = 1, R_HEAVY = 2, R_NEAR = 3, R_FAR = 4 };
constexpr int SSSP_MLP = 4;
__device__ __forceinline__ void relax_range(const uint32_t *__restrict__ tgt, const float *__restrict__ w, uint32_t *dist,
                                            uint32_t *flags, uint32_t *wmin, float du, uint32_t first, uint32_t end,
                                            uint32_t step, uint32_t thr, int which, const uint32_t *__restrict__ final_bits,
                                            RelaxOut &ro, uint32_t cut = NO_BUCKET)
{
    for (uint32_t i = first; i < end; i += step * SSSP_MLP) {
        uint32_t t[SSSP_MLP], nb[SSSP_MLP], pre[SSSP_MLP];
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k) {
            const uint32_t j = i + (uint32_t)k * step;
            const bool in = j < end;
            t[k] = in ? tgt[j] : 0u;
            nb[k] = in ? __float_as_uint(__fadd_rn(du, w[j])) : 0xFFFFFFFFu;
            if (which == R_LIGHT ? nb[k] > thr
                : which == R_HEAVY ? nb[k] <= thr
                : which == R_NEAR  ? (nb[k] <= thr || nb[k] > cut)
                : which == R_FAR   ? nb[k] <= cut
                                   : false)
                nb[k] = 0xFFFFFFFFu;
        }
        if (final_bits) {
            // One bit per node (2 MB at scale 24: it stays in L2) instead of a probe of the 64 MB distance vector, for
            // targets whose distance is known to be final:
            // * heavy round (`settled`): a heavy candidate is beyond the threshold, and a target that was ever taken up had
            //   a distance at or below it — after the phase that takes up most of the graph nearly every target is settled;
            // * light round (`done` = the nodes taken up in EARLIER phases, see sssp_round_kernel): their distances are
            //   final and at or below the previous threshold, every source of the running phase lies beyond it.
            uint32_t bits[SSSP_MLP];
#pragma unroll
            for (int k = 0; k < SSSP_MLP; ++k)
                bits[k] = nb[k] != 0xFFFFFFFFu ? final_bits[t[k] >> 5] : 0u;
#pragma unroll
            for (int k = 0; k < SSSP_MLP; ++k)
                if ((bits[k] >> (t[k] & 31u)) & 1u)
                    nb[k] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            pre[k] = nb[k] != 0xFFFFFFFFu ? ld_agent(&dist[t[k]]) : 0u;
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            relax_checked(dist, flags, wmin, nb[k], pre[k], t[k], thr, ro);
    }
}

// ctrl words shared by the kernels of a round
enum : uint32_t { C_AGAIN = 0, C_FAR = 1, C_BAD = 2, C_THR = 3, C_DONE = 4, C_ROUND = 5, C_ADVANCES = 6,
                  C_WORK = 7 /* edges streamed so far / 64: drives the schedule */, C_HEAVY = 8 /* 1: this round is a phase's heavy round */,
                  C_WIDTH = 9 /* f32 bits: current threshold step */,
                  C_MARK = 10 /* C_WORK at the last advance */, C_TICKET = 11 /* workgroups of sssp_finish_kernel done */,
                  C_SNAP = 12 /* 1: the threshold has just moved: the next round copies `settled` into `done` */,
                  C_NDONE = 13 /* bits set in `done` (as of the last completed snapshot: a lower bound) */,
                  C_NCOUNT = 14 /* ... of the snapshot being taken */,
                  C_CUT = 15 /* f32 bits: heavy rounds relax candidates up to here, the rest waits for the far round; NO_BUCKET: no cut */,
                  C_FAROWED = 16 /* some node has edges waiting for the far round (fflags) */,
                  C_CUTUSED = 17 /* the far round has run: no second cut */,
                  C_WORDS = 32 };
// C_HEAVY: 0 = light round, 1 = a phase's heavy round, 2 = the far round

// The work-item queue of a round is SSSP_QUEUES sub-queues, node group g appending to sub-queue g % SSSP_QUEUES:
// each has its own 64-bit counter (low half: items queued this round, high half: out-edges of the nodes taken up — the
// work statistic rides on the reservation) in its own 128-byte line, and its own stretch of `chunks`, sized once per
// call for the case that every node of its groups is taken up in the same round (sssp_caps_kernel).  One counter
// for the whole grid cost ~15 ns per reservation, serialised: 0.1-0.25 ms of every dense round.
constexpr uint32_t SSSP_QUEUES = 64;
constexpr uint32_t SSSP_QSTRIDE = 16; // counters are 16 x 8 bytes apart
struct QueueState {
    unsigned long long ctr[SSSP_QUEUES * SSSP_QSTRIDE];
    uint32_t start[SSSP_QUEUES]; // first slot of the sub-queue in `chunks`
    uint32_t cap[SSSP_QUEUES];
};

// One round.  wmin[i] is a lower bound of the distances of the flagged nodes of flag word i (32 nodes): whoever
// flags a node lowers it, the scanner resets it and puts back what it leaves flagged.  A wavefront takes 32 words
// (1024 consecutive nodes, 16 per lane) at a time; only words that can hold a node at or below the threshold are
// opened (all at once), their near nodes cleared and collected in an LDS list.  Lists of up to `coop` edges are
// then relaxed inside the wavefront, all their edges, flattened over its lanes.  Everything longer leaves the wavefront: one work item per
// chunk_edges edges in a queue that sssp_chunk_kernel spreads over the whole grid (relaxing the 33..2048-edge lists
// here, one after the other by the whole wavefront, made the few node groups that hold the high-degree nodes — the
// low ids of an RMAT graph — the critical path of every round), and those lists are relaxed in two parts: while
// the phase is busy only the edges that land at or below the threshold, and once, in the phase's heavy round
// (C_HEAVY, the nodes marked in hflags), the others — a hub whose distance improves in ten rounds of a phase has
// its 300 000 targets probed once, not ten times.  Nodes above the threshold cost nothing per round.
constexpr uint32_t SSSP_LANE_NODES = 16u;                // flag bits per lane
constexpr uint32_t SSSP_GROUP = kWave * SSSP_LANE_NODES; // nodes per wavefront step
constexpr uint32_t SSSP_BIG = 2048;  // the work items of a list longer than this are written by the whole wavefront
constexpr uint32_t SSSP_CHUNK = 256; // edges per work item (GM_SSSP_CHUNK overrides)

__device__ __forceinline__ void sssp_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the far round as a PULL ------------------------------------------------------------------------------------------
// When the far round runs, nearly every node that will ever be reached has been taken up; the edges that wait (candidates
// beyond the cut, out of nodes taken up since the cut was set) are a large share of all edges, and pushing them means a
// bit test per edge (235 M of them at RMAT scale 24: 2.3 ms at the L2's random-access rate).  Seen from the other side the
// work is small: only a node that has NOT been taken up can still improve, and such a node takes the minimum of
// dist[s] (+) w over its in-edges — every relaxation that could reach it, waiting or not (extra relaxations are harmless:
// the distances are the least fixed point).  The in-edges come from the transposed copy of the lists in the handle
// (SsspOrder::in_*).  One lane per node (16 nodes per lane, a wavefront per 1024 as everywhere); in-lists longer than
// 256 are read by the whole wavefront.
__device__ __forceinline__ uint32_t sssp_pull_list(const uint2 *__restrict__ in_edge, const uint32_t *dist,
                                                   const uint32_t *settled, uint32_t first, uint32_t end, uint32_t step)
{
    uint32_t best = NO_BUCKET;
    for (uint32_t j = first; j < end; j += step * SSSP_MLP) {
        uint32_t s[SSSP_MLP], ds[SSSP_MLP];
        float wj[SSSP_MLP];
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k) {
            const uint32_t i = j + (uint32_t)k * step;
            const uint2 e = i < end ? in_edge[i] : make_uint2(0u, 0xFFFFFFFFu); // {weight bits, source}: the low / high word of the sort key
            s[k] = e.y;
            wj[k] = __uint_as_float(e.x);
        }
        // GM_SSSP_PULL_FILTER=1 (measured slower: 6.56 against 6.40 ms at scale 24 — nearly every source HAS been taken up by
        // then, the bit test is one more dependent access): only sources that have been taken up matter here (the others
        // relax their edges when their turn comes), so a bit of the L2-resident map could spare the read of the distance
        if (settled) {
            uint32_t sb[SSSP_MLP];
#pragma unroll
            for (int k = 0; k < SSSP_MLP; ++k)
                sb[k] = s[k] != 0xFFFFFFFFu ? ld_agent(&settled[s[k] >> 5]) : 0u;
#pragma unroll
            for (int k = 0; k < SSSP_MLP; ++k)
                if (!((sb[k] >> (s[k] & 31u)) & 1u))
                    s[k] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            ds[k] = s[k] != 0xFFFFFFFFu ? ld_agent(&dist[s[k]]) : SSSP_INF_BITS;
#pragma unroll
        for (int k = 0; k < SSSP_MLP; ++k)
            if (ds[k] != SSSP_INF_BITS) { // sssp.rs:176: an unreached source offers nothing
                const uint32_t nb = __float_as_uint(__fadd_rn(__uint_as_float(ds[k]), wj[k]));
                best = nb < best ? nb : best;
            }
    }
    return best;
}

__device__ __forceinline__ void sssp_pull_apply(uint32_t *dist, uint32_t *flags, uint32_t *wmin, uint32_t t, uint32_t best,
                                                uint32_t thr, RelaxOut &ro)
{
    relax_checked(dist, flags, wmin, best, best != NO_BUCKET ? ld_agent(&dist[t]) : 0u, t, thr, ro);
}

__device__ void sssp_pull_round(const uint32_t *__restrict__ in_off, const uint2 *__restrict__ in_edge, uint32_t *dist,
                                uint32_t *flags, uint32_t *wmin,
                                const uint32_t *settled, uint32_t nwords, uint32_t n, uint32_t thr, RelaxOut &ro,
                                const uint32_t *src_bits)
{
    constexpr uint32_t OWN = 256; // in-edges a lane reads by itself
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t ngroups = (nwords * 2u + kWave - 1) / kWave;
    const uint32_t sh = (lane & 1u) * 16u;
    for (uint32_t grp = wave; grp < ngroups; grp += nwaves) {
        const uint32_t my_word = (grp * kWave + lane) >> 1;
        const uint32_t base = my_word * 32u + sh;
        uint32_t un = my_word < nwords ? ~(ld_agent(&settled[my_word]) >> sh) & 0xFFFFu : 0u; // never taken up
        uint32_t big = 0u;                                                                   // ... with a long in-list
        for (uint32_t bits = un; bits; bits &= bits - 1u) {
            const uint32_t c = (uint32_t)__ffs((int)bits) - 1u, t = base + c;
            if (t >= n)
                break;
            const uint32_t s0 = in_off[t], s1 = in_off[t + 1];
            if (s1 - s0 > OWN) {
                big |= 1u << c;
                continue;
            }
            if (s1 > s0)
                sssp_pull_apply(dist, flags, wmin, t, sssp_pull_list(in_edge, dist, src_bits, s0, s1, 1u), thr, ro);
        }
        uint64_t who;
        while ((who = __ballot(big != 0u)) != 0ull) {
            const int src_lane = __ffsll((unsigned long long)who) - 1;
            const uint32_t bbits = __shfl(big, src_lane, kWave), bbase = __shfl(base, src_lane, kWave);
            const uint32_t c = (uint32_t)__ffs((int)bbits) - 1u, t = bbase + c;
            if ((int)lane == src_lane)
                big &= big - 1u;
            const uint32_t s0 = in_off[t], s1 = in_off[t + 1];
            const uint32_t best = wave_min(sssp_pull_list(in_edge, dist, src_bits, s0 + lane, s1, kWave));
            if (lane == 0)
                sssp_pull_apply(dist, flags, wmin, t, best, thr, ro);
        }
    }
}

// The plan of a handle (SsspOrder), built by two FLAT radix sorts (round 5; until then a segmented sort by weight, 23 ms at
// scale 24, and a transposition by atomics — a histogram of the targets and a cursor per target — 29 ms):
//   key[i] = source << 32 | weight bits of edge i (non-negative floats order like their bit patterns; -0.0 is stored as +0.0,
//   which adds to the same distances)
//   sorted by the whole key with the targets as values  -> every list ordered by weight (targets + the keys' low words)
//   as VALUES of a sort of the targets                   -> the transposed lists: in_edge = {weight bits, source} per in-edge
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_expand_kernel(const uint32_t *__restrict__ off, const float *__restrict__ w, uint32_t n,
                                                                 unsigned long long *__restrict__ key)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + kWave - 1) / kWave * kWave;
    auto make = [](uint32_t r, float x) {
        const uint32_t b = __float_as_uint(x);
        return (unsigned long long)r << 32 | (b == 0x80000000u ? 0u : b);
    };
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_pad; r += stride) {
        const uint32_t s = r < n ? off[r] : 0u, e = r < n ? off[r + 1] : 0u;
        const uint32_t len = e - s;
        if (len <= SSSP_COOP)
            for (uint32_t i = s; i < e; ++i)
                key[i] = make(r, w[i]);
        uint64_t big = __ballot(len > SSSP_COOP);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave), br = __shfl(r, src, kWave);
            for (uint32_t i = bs + lane; i < be; i += kWave)
                key[i] = make(br, w[i]);
        }
    }
}

__global__ void sssp_key_weights_kernel(const unsigned long long *__restrict__ key, uint64_t m, float *__restrict__ w)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        w[i] = __uint_as_float((uint32_t)key[i]);
}

// in_off[t] = first position of the sorted targets with a target >= t (t = 0 .. n), in three steps: every entry = m ("none at or
// behind t"); the first position of every target that occurs; a suffix minimum (rocPRIM scan over reversed iterators).
// (Until round 6 thread i filled in_off for every t in (sorted_tgt[i - 1], sorted_tgt[i]] by itself: a long id range without
// in-edges — the tail of a degree-ordered or partition-local graph — was one thread writing millions of entries, and the
// 32-bit loop bound never ended for n = 2^32 - 1: ADVICE r5.)
__global__ void sssp_fill_u32_kernel(uint32_t *__restrict__ out, uint64_t count, uint32_t value)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        out[i] = value;
}

__global__ void sssp_in_starts_kernel(const uint32_t *__restrict__ sorted_tgt, uint64_t m, uint32_t *__restrict__ in_off)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const uint32_t t = sorted_tgt[i];
        if (i == 0 || sorted_tgt[i - 1] != t)
            in_off[t] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(SSSP_BLOCK) void sssp_round_kernel(const uint32_t *__restrict__ off,
                                                                const uint32_t *__restrict__ tgt,
                                                                const float *__restrict__ w, uint32_t *dist,
                                                                uint32_t *flags, uint32_t *wmin, uint32_t *hflags, uint32_t *fflags,
                                                                uint32_t *settled, uint32_t *done, uint32_t done_min,
                                                                uint32_t nwords, uint2 *__restrict__ chunks,
                                                                QueueState *__restrict__ qs, uint32_t *ctrl,
                                                                uint32_t chunk_edges, uint32_t coop,
                                                                const uint32_t *__restrict__ in_off,
                                                                const uint2 *__restrict__ in_edge, uint32_t n_nodes,
                                                                uint32_t pull_filter)
{
    __shared__ uint16_t list[SSSP_BLOCK / kWave][SSSP_GROUP];          // node - first node of the group
    __shared__ uint8_t owner[SSSP_BLOCK / kWave][kWave * SSSP_COOP];    // short-list edge slot -> lane holding its node
    if (ld_agent(&ctrl[C_DONE]))
        return; // a round enqueued behind the last one of its batch
    const uint32_t thr = ld_agent(&ctrl[C_THR]);
    const uint32_t mode = ld_agent(&ctrl[C_HEAVY]);
    const bool heavy = mode != 0u; // heavy or far round: the nodes come from a bitmap of their own, only long lists
    uint32_t *owed = mode == 2u ? fflags : hflags;
    const bool cut_on = mode == 1u && ld_agent(&ctrl[C_CUT]) != NO_BUCKET; // their edges beyond the cut wait for the far round
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wv = threadIdx.x >> 6;
    // With the pull at hand (it looks at EVERY in-edge of a node that was never taken up) no relaxation beyond the cut has to
    // be made before the far round, whoever the source is: the short lists drop such candidates as well.
    const uint32_t short_cut = in_off && mode == 0u ? ld_agent(&ctrl[C_CUT]) : NO_BUCKET;
    if (mode == 2u && in_off) { // the far round as a pull: nothing is queued, the chunk kernel finds no items
        RelaxOut pro{0u};
        sssp_pull_round(in_off, in_edge, dist, flags, wmin, settled, nwords, n_nodes, thr, pro,
                        pull_filter ? settled : (const uint32_t *)nullptr);
        if (__ballot(pro.again != 0) && lane == 0 && !ld_agent(&ctrl[C_AGAIN]))
            atomicOr(&ctrl[C_AGAIN], 1u);
        return;
    }
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t ngroups = (nwords * 2u + kWave - 1) / kWave;
    const uint32_t sh = (lane & 1u) * 16u; // this lane's half of its flag word
    // `done`: the nodes taken up in earlier phases — final, at or below the previous threshold, so nothing a source of the
    // running phase offers can improve them.  Snapshot of `settled`, taken by the first round after the threshold has
    // moved: every wavefront copies the words of its own groups before it takes up their nodes (a word's bits are only
    // ever set by its owner).  Readers may see the old or the new snapshot of a word: both hold final nodes only.  The
    // bit test replaces the probe of the distance once enough nodes are final to pay for the extra L2 access
    // (done_min; 0xFFFFFFFF = never).
    const bool snap = done != nullptr && ld_agent(&ctrl[C_SNAP]) != 0u;
    const uint32_t *skip_done = done != nullptr && ld_agent(&ctrl[C_NDONE]) >= done_min ? done : nullptr;
    uint32_t copied = 0;
    RelaxOut ro{0u};
    for (uint32_t grp = wave; grp < ngroups; grp += nwaves) {
        const uint32_t my_word = (grp * kWave + lane) >> 1;
        const bool valid = my_word < nwords;
        if (snap && valid && sh == 0u) {
            const uint32_t bits = ld_agent(&settled[my_word]);
            done[my_word] = bits;
            copied += (uint32_t)__popc(bits);
        }
        uint32_t near = 0u; // 16 bits
        if (heavy) {
            near = valid ? (ld_agent(&owed[my_word]) >> sh) & 0xFFFFu : 0u;
            if (!__ballot(near != 0u))
                continue;
            if (near) {
                atomicAnd(&owed[my_word], ~(near << sh));
                if (cut_on) {
                    atomicOr(&fflags[my_word], near << sh);
                    if (!ld_agent(&ctrl[C_FAROWED]))
                        atomicOr(&ctrl[C_FAROWED], 1u);
                }
            }
        } else {
            const uint32_t lo = valid ? ld_agent(&wmin[my_word]) : NO_BUCKET;
            // every lane opens its half word if the word can hold a node at or below the threshold: all candidate
            // words of the group at once — four memory round trips per group whatever the density
            const bool mine = lo <= thr;
            if (!__ballot(mine))
                continue;
            if (mine && sh == 0u)
                atomicExch(&wmin[my_word], NO_BUCKET); // reset first, then look: a concurrent flagger re-arms the word
            sssp_drain();
            const uint32_t fw = mine ? (ld_agent(&flags[my_word]) >> sh) & 0xFFFFu : 0u;
            uint32_t keep_far = NO_BUCKET;
#pragma unroll
            for (uint32_t c = 0; c < SSSP_LANE_NODES; c += 8u) { // the distances of the flagged nodes, eight loads in flight
                uint32_t db[8];
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j)
                    db[j] = ((fw >> (c + j)) & 1u) ? ld_agent(&dist[my_word * 32u + sh + c + j]) : NO_BUCKET;
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j)
                    if ((fw >> (c + j)) & 1u) {
                        if (db[j] <= thr)
                            near |= 1u << (c + j);
                        else
                            keep_far = db[j] < keep_far ? db[j] : keep_far; // what stays flagged goes back into the summary
                    }
            }
            if (keep_far != NO_BUCKET)
                atomicMin(&wmin[my_word], keep_far);
            if (near) {
                atomicAnd(&flags[my_word], ~(near << sh));
                if ((ld_agent(&settled[my_word]) & (near << sh)) != (near << sh))
                    atomicOr(&settled[my_word], near << sh); // taken up = at or below the threshold, now and for ever
            }
            sssp_drain(); // cleared before anyone reads the distances the relaxation uses
        }
        uint32_t cnt = (uint32_t)__popc(near), pre = cnt; // inclusive prefix of the lanes' counts
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t up = __shfl_up(pre, o, kWave);
            if ((int)lane >= o)
                pre += up;
        }
        const uint32_t total = __shfl(pre, kWave - 1, kWave);
        for (uint32_t at = pre - cnt, bits = near; bits; bits &= bits - 1u)
            list[wv][at++] = (uint16_t)(lane * SSSP_LANE_NODES + (uint32_t)__ffs((int)bits) - 1u);
        if (total == 0)
            continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // First the group's item count, so that ONE atomic reserves its stretch of the queue: thousands of
        // wavefronts adding to the same counter once per 64 nodes were ~12 ns apiece, serialised — 0.1-0.4 ms of a
        // dense round.  The same 64-bit add carries the round's work statistic in its upper half.
        uint32_t my_items = 0, my_work = 0;
#pragma unroll 4
        for (uint32_t base = lane; base < total; base += kWave) {
            const uint32_t u = grp * SSSP_GROUP + list[wv][base];
            const uint32_t len = off[u + 1] - off[u];
            my_work += len;
            my_items += len > coop ? (len + chunk_edges - 1u) / chunk_edges : 0u;
        }
        const uint32_t grp_items = (uint32_t)wave_sum((uint64_t)my_items);
        const uint32_t grp_work = (uint32_t)wave_sum((uint64_t)my_work);
        uint32_t cursor = 0;
        if (lane == 0)
            cursor = qs->start[grp % SSSP_QUEUES] +
                     (uint32_t)atomicAdd(&qs->ctr[(grp % SSSP_QUEUES) * SSSP_QSTRIDE],
                                         (unsigned long long)grp_items | ((unsigned long long)grp_work << 32));
        cursor = __shfl(cursor, 0, kWave);
        for (uint32_t base = 0; base < total; base += kWave) {
            uint32_t u = 0, s = 0, e = 0;
            float du = 0.0f;
            if (base + lane < total) {
                u = grp * SSSP_GROUP + list[wv][base + lane];
                du = __uint_as_float(ld_agent(&dist[u])); // <= thr: distances only decrease
                s = off[u];
                e = off[u + 1];
            }
            const uint32_t len = e - s;
            // The short lists of the 64 nodes, flattened over the lanes: edge slot f of their concatenation belongs
            // to the node of lane owner[f].  (One lane walking its own list reads 64 different cache lines per load
            // and takes len / 4 dependent steps; here consecutive lanes read consecutive edges.)
            const uint32_t mine_len = len <= coop ? len : 0u;
            uint32_t sincl = mine_len;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t up = __shfl_up(sincl, o, kWave);
                if ((int)lane >= o)
                    sincl += up;
            }
            const uint32_t slots = __shfl(sincl, kWave - 1, kWave), sfirst = sincl - mine_len;
            if (slots) {
                for (uint32_t j = 0; j < mine_len; ++j)
                    owner[wv][sfirst + j] = (uint8_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t f0 = 0; f0 < slots; f0 += kWave * SSSP_MLP) {
                    uint32_t t[SSSP_MLP], nb[SSSP_MLP], pre[SSSP_MLP];
#pragma unroll
                    for (int k = 0; k < SSSP_MLP; ++k) {
                        const uint32_t f = f0 + (uint32_t)k * kWave + lane;
                        const bool in = f < slots;
                        const int o = in ? (int)owner[wv][f] : 0;
                        const uint32_t j = __shfl(s, o, kWave) + f - __shfl(sfirst, o, kWave);
                        const float od = __shfl(du, o, kWave);
                        t[k] = in ? tgt[j] : 0u;
                        nb[k] = in ? __float_as_uint(__fadd_rn(od, w[j])) : 0xFFFFFFFFu;
                        if (nb[k] > short_cut)
                            nb[k] = 0xFFFFFFFFu;
                    }
                    if (skip_done) {
                        uint32_t bits[SSSP_MLP];
#pragma unroll
                        for (int k = 0; k < SSSP_MLP; ++k)
                            bits[k] = nb[k] != 0xFFFFFFFFu ? skip_done[t[k] >> 5] : 0u;
#pragma unroll
                        for (int k = 0; k < SSSP_MLP; ++k)
                            if ((bits[k] >> (t[k] & 31u)) & 1u)
                                nb[k] = 0xFFFFFFFFu;
                    }
#pragma unroll
                    for (int k = 0; k < SSSP_MLP; ++k)
                        pre[k] = nb[k] != 0xFFFFFFFFu ? ld_agent(&dist[t[k]]) : 0u;
#pragma unroll
                    for (int k = 0; k < SSSP_MLP; ++k)
                        relax_checked(dist, flags, wmin, nb[k], pre[k], t[k], thr, ro);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // owner[] is rewritten by the next step
            }
            const uint32_t nch = len > coop ? (len + chunk_edges - 1u) / chunk_edges : 0u;
            if (!__ballot(nch != 0u))
                continue;
            if (nch && !heavy)
                atomicOr(&hflags[u >> 5], 1u << (u & 31u)); // owed a heavy round when the phase has run dry
            uint32_t incl = nch;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o, kWave);
                if ((int)lane >= o)
                    incl += up;
            }
            const uint32_t first = cursor + incl - nch;
            cursor += __shfl(incl, kWave - 1, kWave);
            if (len <= SSSP_BIG)
                for (uint32_t c = 0; c < nch; ++c)
                    chunks[first + c] = make_uint2(u, s + c * chunk_edges);
            // hubs: the item list of one node written by the whole wavefront
            uint64_t huge = __ballot(len > SSSP_BIG);
            while (huge) {
                const int src = __ffsll((unsigned long long)huge) - 1;
                huge &= huge - 1;
                const uint32_t hu = __shfl(u, src, kWave), hs = __shfl(s, src, kWave);
                const uint32_t hn = __shfl(nch, src, kWave), hfirst = __shfl(first, src, kWave);
                for (uint32_t c = lane; c < hn; c += kWave)
                    chunks[hfirst + c] = make_uint2(hu, hs + c * chunk_edges);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the list is rewritten by the next step
    }
    if (snap && (wave & 63u) == 0u) {
        // an estimate from every 64th wavefront (their groups are spread evenly over the ids): 16 000 wavefronts adding to
        // one word serialise at ~15 ns apiece — 0.25 ms per snapshot, 6 ms of a 9 ms call at scale 24 when all of them did
        copied = (uint32_t)wave_sum((uint64_t)copied);
        if (lane == 0 && copied)
            atomicAdd(&ctrl[C_NCOUNT], copied * 64u);
    }
    if (__ballot(ro.again != 0) && lane == 0 && !ld_agent(&ctrl[C_AGAIN]))
        atomicOr(&ctrl[C_AGAIN], 1u);
}

// The work items of the round that just ran: item (u, first edge) -> up to chunk_edges edges of u, one wavefront per
// item, any wavefront of the grid; in a light round only the edges that land at or below the threshold are probed,
// in a heavy round only the others.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_chunk_kernel(const uint32_t *__restrict__ off,
                                                                const uint32_t *__restrict__ tgt,
                                                                const float *__restrict__ w, uint32_t *dist,
                                                                uint32_t *flags, uint32_t *wmin,
                                                                const uint32_t *__restrict__ settled,
                                                                const uint32_t *__restrict__ done, uint32_t done_min,
                                                                const uint2 *__restrict__ chunks,
                                                                const QueueState *__restrict__ qs, uint32_t *ctrl,
                                                                uint32_t chunk_edges, const uint32_t *__restrict__ tgt_by_w,
                                                                const float *__restrict__ w_by_w)
{
    if (ld_agent(&ctrl[C_DONE]))
        return;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    // lane q looks at sub-queue q: the wavefront walks the concatenation of the 64 sub-queues
    static_assert(SSSP_QUEUES == kWave, "one sub-queue per lane");
    const uint32_t q_items = (uint32_t)qs->ctr[lane * SSSP_QSTRIDE]; // written by the round kernel before this launch
    uint32_t q_incl = q_items;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t up = __shfl_up(q_incl, o, kWave);
        if ((int)lane >= o)
            q_incl += up;
    }
    const uint32_t nchunks = __shfl(q_incl, kWave - 1, kWave);
    if (nchunks == 0)
        return;
    const uint32_t q_base = qs->start[lane] - (q_incl - q_items); // slot of item f of sub-queue q = q_base + f
    const uint32_t thr = ld_agent(&ctrl[C_THR]);
    const uint32_t mode = ld_agent(&ctrl[C_HEAVY]);
    const bool heavy = mode != 0u;
    const uint32_t cut = ld_agent(&ctrl[C_CUT]);
    // heavy round: targets ever taken up; light round: targets taken up in earlier phases (once there are enough of them)
    const uint32_t *final_bits = heavy ? settled : (done && ld_agent(&ctrl[C_NDONE]) >= done_min ? done : nullptr);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    RelaxOut ro{0u};
    // (Testing 64 items at once, one per lane, and relaxing the survivors one after the other was measured and dropped:
    // the test got cheaper, but a hub's light items — consecutive in the queue — then ran one behind the other in a few
    // wavefronts, improvements spread more slowly and the phase needed more passes: 2.94 -> 3.20 x m edges, 5.56 -> 5.75 ms.)
    for (uint32_t c = wave; c < nchunks; c += nwaves) {
        const int q = __popcll(__ballot(q_incl <= c)); // sub-queues that end at or before item c
        const uint2 ch = chunks[__shfl(q_base, q, kWave) + c];
        const float du = __uint_as_float(ld_agent(&dist[ch.x]));
        const uint32_t end_u = off[ch.x + 1];
        const uint32_t end = ch.y + chunk_edges < end_u ? ch.y + chunk_edges : end_u;
        // Light rounds read the lists ordered by weight (SsspOrder): fl(du + w) does not decrease along such a list, so the
        // light edges are a prefix — an item whose first edge already lands beyond the threshold has nothing to do: two loads
        // instead of 256 edges.  The heavy round (nearly every edge of the list) reads the CSR's own, target-ordered lists:
        // consecutive edges of a hub probe neighbouring words of `settled` (14 lines per item instead of 256).
        if (!heavy && w_by_w) {
            if (__float_as_uint(__fadd_rn(du, w_by_w[ch.y])) > thr)
                continue;
            relax_range(tgt_by_w, w_by_w, dist, flags, wmin, du, ch.y + lane, end, kWave, thr, R_LIGHT, final_bits, ro);
            continue;
        }
        // A cut above the threshold (sssp_advance sets it once, at the end of the first large phase): the heavy round relaxes
        // only the candidates up to the cut — a slice of the weight-ordered list, items outside it dropped after two loads —
        // and the others wait for ONE far round, run when the threshold is about to pass the cut.  By then most targets have
        // been taken up and are skipped by their bit in `settled` (2 MB, L2) instead of a probe of the 64 MB distance vector:
        // the heavy round of the large phase had 130 M L2 misses for its 142 M edges (rocprofv3 TCC_MISS, scale 24).
        if (mode == 1u && cut != NO_BUCKET && w_by_w) {
            if (__float_as_uint(__fadd_rn(du, w_by_w[ch.y])) > cut || __float_as_uint(__fadd_rn(du, w_by_w[end - 1u])) <= thr)
                continue;
            relax_range(tgt_by_w, w_by_w, dist, flags, wmin, du, ch.y + lane, end, kWave, thr, R_NEAR, final_bits, ro, cut);
            continue;
        }
        relax_range(tgt, w, dist, flags, wmin, du, ch.y + lane, end, kWave, thr, mode == 2u ? R_FAR : heavy ? R_HEAVY : R_LIGHT,
                    final_bits, ro, cut);
    }
    if (__ballot(ro.again != 0) && lane == 0 && !ld_agent(&ctrl[C_AGAIN]))
        atomicOr(&ctrl[C_AGAIN], 1u);
}

// Once per call: the capacity of every sub-queue = the items its node groups queue when all their nodes are taken
// up in one round, then the sub-queues' first slots.
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_caps_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t ngroups,
                                                               QueueState *__restrict__ qs, uint32_t chunk_edges, uint32_t coop)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t grp = wave; grp < ngroups; grp += nwaves) {
        uint32_t items = 0;
        const uint32_t u0 = grp * SSSP_GROUP + lane * SSSP_LANE_NODES;
        for (uint32_t u = u0; u < u0 + SSSP_LANE_NODES && u < n; ++u) {
            const uint32_t len = off[u + 1] - off[u];
            items += len > coop ? (len + chunk_edges - 1u) / chunk_edges : 0u;
        }
        items = (uint32_t)wave_sum((uint64_t)items);
        if (lane == 0 && items)
            atomicAdd(&qs->cap[grp % SSSP_QUEUES], items);
    }
}

__global__ void sssp_qstart_kernel(QueueState *qs)
{
    uint32_t at = 0;
    for (uint32_t q = 0; q < SSSP_QUEUES; ++q) {
        qs->start[q] = at;
        at += qs->cap[q];
    }
}

// Between two rounds (one wavefront): nothing left below the threshold -> move it to the minimum pending
// distance + width, or finish; then clear the round's outputs.
// adapt_lo / adapt_hi (units of 64 relaxed edges; 0 = fixed width): the step halves when the phase that
// just ended relaxed more than adapt_hi and doubles when it relaxed less than adapt_lo, within
// [width_min, width_max] — coarse steps re-relax every edge ~6 times, fine steps leave the chip idle.
struct AdvanceParams {
    uint32_t adapt_lo, adapt_hi; // units of 64 streamed edges; 0 = fixed width
    float width_min, width_max;
    float cut_mult;    // the cut = threshold + cut_mult x width at the end of the first large phase; 0: no cut
    uint32_t cut_work; // "large": the phase streamed at least this much (units of 64 edges)
    uint32_t pull;     // the far round pulls: the cut may be set as soon as a phase turns out large, and binds every source
};
__device__ void sssp_advance(uint32_t *ctrl, QueueState *qs, const AdvanceParams p)
{
    // 64 threads: thread q reads and clears sub-queue q's counter; thread 0 does the rest
    const uint64_t round_work = wave_sum((uint64_t)(qs->ctr[threadIdx.x * SSSP_QSTRIDE] >> 32));
    qs->ctr[threadIdx.x * SSSP_QSTRIDE] = 0ull;
    if (threadIdx.x != 0)
        return;
    ctrl[C_WORK] += (uint32_t)((round_work + 63u) >> 6);
    if (ctrl[C_SNAP]) { // the round that just ran took the snapshot: publish its count
        ctrl[C_NDONE] = ld_agent(&ctrl[C_NCOUNT]);
        ctrl[C_NCOUNT] = 0u;
    }
    ctrl[C_SNAP] = 0u;
    const uint32_t far = ld_agent(&ctrl[C_FAR]); // folded in by other workgroups of this launch: not through L1
    const uint32_t mode = ctrl[C_HEAVY];
    if (mode == 0u) {
        const bool dry = !ctrl[C_AGAIN];
        if (dry)
            ctrl[C_HEAVY] = 1u; // the phase has run dry: next, the heavy round of the nodes it took up
        // ... which, from the first large phase on, stops at a cut above the threshold (sssp_chunk_kernel): once per call.
        // With the pull the cut is set as soon as the phase has turned out large, in the middle of it.
        if ((dry || p.pull) && p.cut_mult > 0.0f && ctrl[C_CUT] == NO_BUCKET && !ctrl[C_CUTUSED] &&
            ctrl[C_WORK] - ctrl[C_MARK] >= p.cut_work) {
            const float c = __fadd_rn(__uint_as_float(ctrl[C_THR]), p.cut_mult * __uint_as_float(ctrl[C_WIDTH]));
            if (c < 3.0e38f && __float_as_uint(c) > ctrl[C_THR]) {
                ctrl[C_CUT] = __float_as_uint(c);
                if (p.pull)
                    ctrl[C_FAROWED] = 1u; // (without the pull: set by the heavy rounds, for the nodes in fflags)
            }
        }
    } else {
        const bool after_far = mode == 2u;
        if (after_far) { // every waiting edge has been relaxed: from here on heavy rounds are whole again
            ctrl[C_FAROWED] = 0u;
            ctrl[C_CUT] = NO_BUCKET;
            ctrl[C_CUTUSED] = 1u;
        }
        ctrl[C_HEAVY] = 0u;
        const bool owed = ctrl[C_FAROWED] != 0u;
        if (ctrl[C_AGAIN]) {
            // cannot happen (a heavy round only writes distances beyond the threshold); if it did, the phase goes on
        } else if (far == NO_BUCKET) {
            if (owed)
                ctrl[C_HEAVY] = 2u; // nothing pending, but edges are: the far round may still find nodes
            else
                ctrl[C_DONE] = 1u;
        } else {
            float width = __uint_as_float(ctrl[C_WIDTH]);
            if (p.adapt_hi && !after_far) { // (the far round is not a phase: the step was adapted when the phase before it ended)
                const uint32_t phase = ctrl[C_WORK] - ctrl[C_MARK];
                if (phase > p.adapt_hi)
                    width = fmaxf(width * 0.5f, p.width_min);
                else if (phase < p.adapt_lo)
                    width = fminf(width * 2.0f, p.width_max);
            }
            const float next = __fadd_rn(__uint_as_float(far), width);
            uint32_t nb = __float_as_uint(next);
            nb = nb > far && next < 3.0e38f ? nb : far; // always covers the pending minimum
            if (p.adapt_hi) {
                ctrl[C_WIDTH] = __float_as_uint(width);
                ctrl[C_MARK] = ctrl[C_WORK];
            }
            if (owed && nb > ctrl[C_CUT]) {
                ctrl[C_HEAVY] = 2u; // the threshold is about to pass the cut: first the edges that wait beyond it
            } else {
                ctrl[C_THR] = nb > ctrl[C_THR] ? nb : ctrl[C_THR]; // (a stale-low word bound never moves it back)
                ctrl[C_ADVANCES] += 1u;
                ctrl[C_SNAP] = 1u; // the next round snapshots `settled` into `done` and counts it
            }
        }
    }
    ctrl[C_AGAIN] = 0u;
    ctrl[C_FAR] = NO_BUCKET;
    ctrl[C_ROUND] += 1u;
}

// The tail of a round, one launch: after a phase's heavy round every workgroup folds its share of the word bounds into
// the pending minimum (every flagged node has its word's bound at or below its distance once the round's atomics have
// landed; a bound may be stale-low, the next round then opens that word and tightens it) and the last one to finish
// moves the threshold; after a light round workgroup 0 does the bookkeeping alone.  (Thousands of wavefronts each
// folding their own minimum into one ctrl word was ~0.1 ms of every round; a separate launch for the bookkeeping ~4 us.)
__global__ __launch_bounds__(SSSP_BLOCK) void sssp_finish_kernel(const uint32_t *__restrict__ wmin, uint32_t nwords,
                                                                 uint32_t *ctrl, QueueState *qs, const AdvanceParams p)
{
    __shared__ uint32_t part[SSSP_BLOCK / kWave];
    __shared__ uint32_t last;
    if (ld_agent(&ctrl[C_DONE]))
        return;
    if (!ld_agent(&ctrl[C_HEAVY])) {
        if (blockIdx.x == 0 && threadIdx.x < kWave)
            sssp_advance(ctrl, qs, p);
        return;
    }
    uint32_t lo = NO_BUCKET;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t v = ld_agent(&wmin[i]);
        lo = v < lo ? v : lo;
    }
    lo = wave_min(lo);
    if ((threadIdx.x & (kWave - 1)) == 0)
        part[threadIdx.x >> 6] = lo;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < SSSP_BLOCK / kWave; ++k)
            lo = part[k] < lo ? part[k] : lo;
        if (lo != NO_BUCKET)
            atomicMin(&ctrl[C_FAR], lo);
        __threadfence();
        last = atomicAdd(&ctrl[C_TICKET], 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!last)
        return;
    if (threadIdx.x == 0)
        st_agent(&ctrl[C_TICKET], 0u);
    __threadfence();
    if (threadIdx.x < kWave)
        sssp_advance(ctrl, qs, p);
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
    const bool times = getenv("GM_SSSP_TIMES") != nullptr; // where the wall time of the call goes, on stderr
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    const uint32_t n = (uint32_t)g->n;
    const uint32_t nwords = (n + 31u) / 32u; // one flag bit per node, one distance summary per flag word
    uint32_t chunk_edges = SSSP_CHUNK;
    if (const char *v = getenv("GM_SSSP_CHUNK"))
        if (atoi(v) >= 64)
            chunk_edges = (uint32_t)atoi(v);
    uint32_t coop = SSSP_COOP;
    if (const char *v = getenv("GM_SSSP_COOP"))
        if (atoi(v) >= 1 && atoi(v) <= (int)SSSP_COOP)
            coop = (uint32_t)atoi(v);
    // GM_SSSP_ARENA=<mask>: which buffers come from the arena — 1 the call's scratch, 2 the kept weight-ordered lists, 4 / 8 the
    // transposed lists (in_edge / in_off) allocated where the transposition starts, 16 the transposed lists allocated BEFORE the
    // build's first temporary is released.  Default 19 = 1 + 2 + 16: no hipMalloc in a plan build (its 4 GB at scale 24 waited
    // for the driver to clear what other processes had freed: 161 ms instead of 41 on the driver's box in round 5).
    // Why 16 and not 4 + 8: with BOTH transposed lists mapped at the transposition's start — into the pieces and the address
    // ranges the build's first half has just released (LIFO, exact-size reuse) — the second sort dies of a GPU memory fault, every
    // time (9 runs of 9: tools/runs/r06_call04.sh, r06_call07.sh, r06_call10.sh; either one alone 2 of 2 passed; allocated up
    // front 4 of 4 passed, bit-exact).  Every access of both buffers is bounds-checked and the same code on hipMalloc'd buffers
    // has never faulted: a runtime problem with that unmap / map sequence, not understood further.
    const int arena_mask = gm::measure_env("GM_SSSP_ARENA") ? atoi(gm::measure_env("GM_SSSP_ARENA")) : 19;
    std::unique_ptr<gm::SsspScratch> sc;
    {
        std::lock_guard<std::mutex> lock(g->cache_mu);
        sc = std::move(g->sssp_scratch);
    }
    struct Park { // hand the buffers back to the handle on every way out
        const gm_csr *g;
        std::unique_ptr<gm::SsspScratch> &sc;
        ~Park()
        {
            std::lock_guard<std::mutex> lock(g->cache_mu);
            if (sc && sc->items && !g->sssp_scratch)
                g->sssp_scratch = std::move(sc);
        }
    } park{g, sc};
    // (node, first edge) work items of one round: a node is taken up at most once per round, so at most one item
    // per chunk_edges edges plus one per list that is not relaxed by its own lane
    const size_t items = (size_t)g->m / chunk_edges + (size_t)g->m / (coop + 1u) + 64;
    if (!sc || sc->items < items) {
        sc.reset(new gm::SsspScratch);
        // (the two large ones from the arena's idle pieces where it has them: a hipMalloc behind another process's — or this
        //  one's — hipFree waits for the driver to clear the freed memory, 8 ms of a first call at scale 24)
        GM_TRY((arena_mask & 1) ? sc->dist.alloc_scratch((size_t)n * 4) : sc->dist.alloc((size_t)n * 4));
        GM_TRY(sc->flags.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->wmin.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->hflags.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->settled.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->done.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->ctrl.alloc(C_WORDS * 4));
        GM_TRY(sc->fflags.alloc(((size_t)nwords + kWave) * 4));
        GM_TRY(sc->queues.alloc(sizeof(QueueState)));
        GM_TRY(sc->queues_init.alloc(sizeof(QueueState)));
        sc->caps_key = 0;
        GM_TRY(sc->hctrl.alloc(C_WORDS * 4));
        GM_TRY((arena_mask & 1) ? sc->chunks.alloc_scratch(items * sizeof(uint2)) : sc->chunks.alloc(items * sizeof(uint2)));
        sc->items = items;
        if (times)
            fprintf(stderr, "sssp: the call's buffers allocated after %.3f ms\n", since(t_call));
    }
    // The lists once more, ordered by weight and transposed (~20 B per edge, released by gm_csr_trim):
    // a long list is relaxed in two parts — while its phase is busy only the edges that land at or below the threshold,
    // once, when the phase has run dry, the others — and with the thresholds a graph of this kind needs (distances of a
    // few hundredths under weights uniform in (0, 1]) the first part is a few per cent of the list.  In target order the
    // whole list is streamed every time its node is taken up (2.9 x m edges per call at RMAT scale 24).  Any order of a
    // list gives the same distances (the least fixed point does not depend on the schedule).  GM_SSSP_ORDER=0: CSR order, 1: also below 2^20 edges.
    std::shared_ptr<const gm::SsspOrder> order;
    const char *ord_env = getenv("GM_SSSP_ORDER");
    // Built by the SECOND call on a handle (the reference's app calls delta_stepping in a loop: app.rs:124-153): one call
    // alone is faster without (scale 24: 9.2 ms against 60 + 6.4), a loop pays the 60 ms back after twenty calls... and a
    // first call that is quick keeps one-shot users where they were.  GM_SSSP_ORDER=1: build at once, whatever the size.
    const uint64_t calls_before = g->sssp_calls.fetch_add(1, std::memory_order_relaxed);
    if (g->m && (ord_env ? atoi(ord_env) != 0 : g->m >= ((uint64_t)1 << 20) && calls_before >= 1)) { // (small graphs: launch-bound)
        {
            std::lock_guard<std::mutex> lock(g->cache_mu);
            order = g->sssp_order;
        }
        std::unique_lock<std::mutex> build_lock(g->sssp_build_mu, std::defer_lock);
        if (!order && !g->sssp_order_failed.load(std::memory_order_relaxed)) {
            build_lock.lock(); // one builder per handle: a concurrent second call waits here and finds the lists built
            std::lock_guard<std::mutex> lock(g->cache_mu);
            order = g->sssp_order;
        }
        // Room: the lists keep 16 B per edge + 4 B per node; the two sorts hold about 36 B per edge at their peak.  The build
        // only starts when the device has that, and a quarter more, FREE (another plan or the caller's tensors may need the
        // rest); a build that did not happen is remembered in the handle (until gm_csr_trim) instead of retried by every call.
        if (!order && build_lock.owns_lock() && !g->sssp_order_failed.load(std::memory_order_relaxed)) {
            size_t free_b = 0, total_b = 0;
            const size_t need = (size_t)g->m * ((arena_mask & 16) ? 44 : 36) + ((size_t)n + 1) * 4; // (the transposed lists up front: 8 B per edge earlier)
            uint64_t arena_b[4] = {0, 0, 0, 0}; // the arena's idle pieces do not show up as free memory, and serve every buffer below
            if (gm::arena_enabled())
                gm::arena_stats(g->device, arena_b);
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b + arena_b[1] < need + need / 4) {
                (void)hipGetLastError();
                g->sssp_order_failed.store(1, std::memory_order_relaxed);
                if (gm::log_enabled())
                    fprintf(stderr, "[graph_mi355x] sssp: %zu MiB free, the weight-ordered lists need %zu MiB while they are built: "
                                    "running on the CSR's own\n", free_b >> 20, (need + need / 4) >> 20);
            }
        }
        if (!order && build_lock.owns_lock() && !g->sssp_order_failed.load(std::memory_order_relaxed)) {
            // out of memory while building them is not an error of the call: it runs on the CSR's lists, as the first call did
            int key_bits = 0;
            while (((uint64_t)1 << key_bits) < n)
                ++key_bits;
            gm::DevBuf key; // source << 32 | weight bits of every edge, CSR order: the key of one sort, the values of the other
            auto build = [&](std::shared_ptr<gm::SsspOrder> &fresh) -> int {
                gm::DevBuf key_sorted, temp;
                // (the kept lists from the arena as well, round 6: the second call took 40 ms on a box that had run nothing
                //  else and 160 ms behind other processes' frees — on the driver's box and here alike: hipMalloc of 4 GB waiting
                //  for the driver to clear what others had freed)
                GM_TRY((arena_mask & 2) ? fresh->targets.alloc_big((size_t)g->m * 4) : fresh->targets.alloc((size_t)g->m * 4));
                GM_TRY((arena_mask & 2) ? fresh->weights.alloc_big((size_t)g->m * 4) : fresh->weights.alloc((size_t)g->m * 4));
                // GM_SSSP_ARENA bit 16: the transposed lists' buffers NOW, before any temporary of this build has been released —
                // an arena buffer that takes over the address range a temporary has just given up (exact-size reuse: in_edge
                // and key_sorted are both 8 B per edge) is what the fault of masks 4 / 8 looks like it needs
                if ((arena_mask & 16) && !(getenv("GM_SSSP_PULL") && atoi(getenv("GM_SSSP_PULL")) == 0)) {
                    GM_TRY(fresh->in_off.alloc_scratch(((size_t)n + 1) * 4));
                    GM_TRY(fresh->in_edge.alloc_big((size_t)g->m * 8));
                }
                GM_TRY(key.alloc_scratch((size_t)g->m * 8));
                GM_TRY(key_sorted.alloc_scratch((size_t)g->m * 8));
                unsigned eg = gm::div_up(n, SSSP_BLOCK);
                hipLaunchKernelGGL(sssp_expand_kernel, dim3(eg > 8192 ? 8192 : eg), dim3(SSSP_BLOCK), 0, (hipStream_t)0, g->offsets,
                                   g->weights, n, key.as<unsigned long long>());
                size_t temp_bytes = 0;
                GM_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, key.as<unsigned long long>(), key_sorted.as<unsigned long long>(),
                                                 g->targets, fresh->targets.as<uint32_t>(), (size_t)g->m, 0u, 32u + (unsigned)key_bits,
                                                 (hipStream_t)0));
                GM_TRY(temp.alloc_scratch(temp_bytes ? temp_bytes : 4));
                GM_HIP(rocprim::radix_sort_pairs(temp.p, temp_bytes, key.as<unsigned long long>(), key_sorted.as<unsigned long long>(),
                                                 g->targets, fresh->targets.as<uint32_t>(), (size_t)g->m, 0u, 32u + (unsigned)key_bits,
                                                 (hipStream_t)0));
                unsigned wg = gm::div_up(g->m, 256);
                hipLaunchKernelGGL(sssp_key_weights_kernel, dim3(wg > 16384 ? 16384 : wg), dim3(256), 0, (hipStream_t)0,
                                   key_sorted.as<unsigned long long>(), g->m, fresh->weights.as<float>());
                GM_HIP(hipGetLastError());
                GM_HIP(hipStreamSynchronize((hipStream_t)0));
                if (times)
                    fprintf(stderr, "sssp: lists sorted by (node, weight) after %.3f ms\n", since(t_call));
                return GM_OK;
            };
            auto transpose = [&](std::shared_ptr<gm::SsspOrder> &fresh) -> int { // the in-edges for the far round's pull
                gm::DevBuf tgt_sorted, temp;
                if (!fresh->in_off.p)
                    GM_TRY((arena_mask & 8) ? fresh->in_off.alloc_scratch(((size_t)n + 1) * 4) : fresh->in_off.alloc(((size_t)n + 1) * 4));
                if (!fresh->in_edge.p)
                    GM_TRY((arena_mask & 4) ? fresh->in_edge.alloc_big((size_t)g->m * 8) : fresh->in_edge.alloc((size_t)g->m * 8));
                GM_TRY(tgt_sorted.alloc_scratch((size_t)g->m * 4));
                size_t temp_bytes = 0;
                GM_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, g->targets, tgt_sorted.as<uint32_t>(), key.as<unsigned long long>(),
                                                 fresh->in_edge.as<unsigned long long>(), (size_t)g->m, 0u, (unsigned)(key_bits ? key_bits : 1),
                                                 (hipStream_t)0));
                GM_TRY(temp.alloc_scratch(temp_bytes ? temp_bytes : 4));
                GM_HIP(rocprim::radix_sort_pairs(temp.p, temp_bytes, g->targets, tgt_sorted.as<uint32_t>(), key.as<unsigned long long>(),
                                                 fresh->in_edge.as<unsigned long long>(), (size_t)g->m, 0u, (unsigned)(key_bits ? key_bits : 1),
                                                 (hipStream_t)0));
                unsigned fg = gm::div_up((uint64_t)n + 1, 256), bg = gm::div_up(g->m, 256);
                uint32_t *in_off = fresh->in_off.as<uint32_t>();
                hipLaunchKernelGGL(sssp_fill_u32_kernel, dim3(fg > 16384 ? 16384 : fg), dim3(256), 0, (hipStream_t)0, in_off,
                                   (uint64_t)n + 1, (uint32_t)g->m);
                hipLaunchKernelGGL(sssp_in_starts_kernel, dim3(bg > 16384 ? 16384 : bg), dim3(256), 0, (hipStream_t)0,
                                   tgt_sorted.as<uint32_t>(), g->m, in_off);
                GM_HIP(hipGetLastError());
                {
                    auto rev = rocprim::make_reverse_iterator(in_off + (size_t)n + 1); // from in_off[n] down to in_off[0]
                    size_t scan_bytes = 0;
                    GM_HIP(rocprim::inclusive_scan(nullptr, scan_bytes, rev, rev, (size_t)n + 1, rocprim::minimum<uint32_t>(), (hipStream_t)0));
                    if (scan_bytes > temp.bytes)
                        GM_TRY(temp.alloc_scratch(scan_bytes));
                    GM_HIP(rocprim::inclusive_scan(temp.p, scan_bytes, rev, rev, (size_t)n + 1, rocprim::minimum<uint32_t>(), (hipStream_t)0));
                }
                GM_HIP(hipStreamSynchronize((hipStream_t)0));
                return GM_OK;
            };
            auto fresh = std::make_shared<gm::SsspOrder>();
            if (build(fresh) == GM_OK) {
                if (!(getenv("GM_SSSP_PULL") && atoi(getenv("GM_SSSP_PULL")) == 0) && transpose(fresh) != GM_OK) {
                    (void)hipGetLastError();
                    fresh->in_off.release(); // the far round pushes
                    fresh->in_edge.release();
                }
                if (times)
                    fprintf(stderr, "sssp: lists ordered by weight%s in %.3f ms (kept in the handle)\n",
                            fresh->in_off.p ? " and transposed" : "", since(t_call));
                std::lock_guard<std::mutex> lock(g->cache_mu);
                if (!g->sssp_order)
                    g->sssp_order = fresh;
                order = g->sssp_order;
            } else {
                (void)hipGetLastError();
                g->sssp_order_failed.store(1, std::memory_order_relaxed);
                if (gm::log_enabled())
                    fprintf(stderr, "[graph_mi355x] sssp: no room for the weight-ordered lists (%s): running on the CSR's own\n",
                            gm_last_error());
            }
        }
    }
    const uint32_t *e_tgt = order ? order->targets.as<uint32_t>() : g->targets;
    const float *e_w = order ? order->weights.as<float>() : g->weights;
    gm::DevBuf &dist = sc->dist, &flags = sc->flags, &wmin = sc->wmin, &hflags = sc->hflags, &settled = sc->settled,
               &ctrl = sc->ctrl,
               &chunks = sc->chunks;
    QueueState *qs = sc->queues.as<QueueState>();
    gm::PinnedBuf &hctrl = sc->hctrl;
    hipStream_t st = 0;
    unsigned grid = gm::div_up(n, SSSP_BLOCK);
    if (grid > 256 * 8)
        grid = 256 * 8;
    const uint32_t ngroups = gm::div_up((uint64_t)nwords * 2u, (uint64_t)kWave);
    unsigned round_grid = gm::div_up(ngroups, SSSP_BLOCK / kWave); // one node group per wavefront
    round_grid = round_grid > 256 * 16 ? 256 * 16 : round_grid;
    // Threshold step: starts at delta/32 and adapts to the work of each phase (sssp_advance): it doubles
    // while a phase streams fewer than 3m/4 edges (without an upper limit: on a long path with weights far above
    // delta a capped step would move the threshold one node at a time) and halves beyond 3m.  Measured at RMAT scale 24, delta 0.1:
    // 2.0 x m relaxations in ~60 rounds, 32 ms; a fixed step of delta: 6.4 x m, 53 ms; fixed delta/16: 2.2 x m but
    // 500 rounds, 101 ms.  GM_SSSP_WIDTH=<fraction of delta> sets the first step, GM_SSSP_ADAPT="lo,hi" the band
    // in millions of edges ("0,0": fixed step).
    float frac = 1.0f / 32.0f;
    if (const char *v = getenv("GM_SSSP_WIDTH"))
        frac = (float)atof(v);
    if (!(frac > 0.0f))
        frac = 1.0f / 32.0f;
    const float width = delta * frac;
    // (work = edges streamed by the phase, light re-streams included: 0.75 m .. 3 m measured best at scale 24,
    //  9.2 vs 9.4 ms for m/5 .. 3m/4, the band of the round kernels that probed every streamed edge)
    uint32_t adapt_lo = (uint32_t)(g->m / 4 * 3 / 64), adapt_hi = (uint32_t)((g->m / 64) * 3) + 1u;
    if (const char *v = getenv("GM_SSSP_ADAPT")) {
        double lo = 0, hi = 0;
        if (sscanf(v, "%lf,%lf", &lo, &hi) == 2 && lo >= 0) {
            adapt_lo = hi > lo ? (uint32_t)(lo * 1e6 / 64.0) : 0u;
            adapt_hi = hi > lo ? (uint32_t)(hi * 1e6 / 64.0) : 0u;
        }
    }

    // ctrl: again 0, far NONE, bad 0, threshold 0.0 (only the start node qualifies), done 0, round 0, advances 0
    uint32_t init_ctrl[C_WORDS] = {0u, NO_BUCKET};
    init_ctrl[C_CUT] = NO_BUCKET;
    memcpy(&init_ctrl[C_WIDTH], &width, 4);
    GM_HIP(hipMemcpyAsync(ctrl.p, init_ctrl, C_WORDS * 4, hipMemcpyHostToDevice, st));
    // the weights of a handle do not change: one look per handle (0.3 ms of a 10 ms call at scale 24)
    const bool check_weights = g->m && g->weights_ok.load(std::memory_order_relaxed) == 0;
    if (check_weights) {
        unsigned wg = gm::div_up(g->m, 256);
        hipLaunchKernelGGL(sssp_check_weights_kernel, dim3(wg > 8192 ? 8192 : wg), dim3(256), 0, st, g->weights, g->m,
                           ctrl.as<uint32_t>() + C_BAD);
    }
    hipLaunchKernelGGL(sssp_init_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, dist.as<uint32_t>(), n,
                       (uint32_t)start_node);
    GM_HIP(hipMemsetAsync(flags.p, 0, flags.bytes, st));
    GM_HIP(hipMemsetAsync(hflags.p, 0, hflags.bytes, st));
    GM_HIP(hipMemsetAsync(sc->fflags.p, 0, sc->fflags.bytes, st));
    GM_HIP(hipMemsetAsync(settled.p, 0, settled.bytes, st));
    GM_HIP(hipMemsetAsync(sc->done.p, 0, sc->done.bytes, st));
    // the sub-queues' capacities and first slots depend on the graph and the work split only: once per parked scratch
    const uint64_t caps_key = ((uint64_t)chunk_edges << 32 | coop) + 1u;
    if (sc->caps_key != caps_key) {
        QueueState *qi = sc->queues_init.as<QueueState>();
        GM_HIP(hipMemsetAsync(qi, 0, sizeof(QueueState), st));
        hipLaunchKernelGGL(sssp_caps_kernel, dim3(round_grid), dim3(SSSP_BLOCK), 0, st, g->offsets, n, ngroups, qi, chunk_edges,
                           coop);
        hipLaunchKernelGGL(sssp_qstart_kernel, dim3(1), dim3(1), 0, st, qi);
        sc->caps_key = caps_key;
    }
    GM_HIP(hipMemcpyAsync(qs, sc->queues_init.p, sizeof(QueueState), hipMemcpyDeviceToDevice, st));
    const uint32_t start_bit = 1u << (start_node & 31u);
    GM_HIP(hipMemcpyAsync(flags.as<uint32_t>() + (start_node >> 5), &start_bit, 4, hipMemcpyHostToDevice, st));
    GM_HIP(hipMemsetAsync(wmin.p, 0xFF, wmin.bytes, st));
    GM_HIP(hipMemsetAsync(wmin.as<uint32_t>() + (start_node >> 5), 0, 4, st)); // the start node's distance: 0.0
    GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, C_WORDS * 4, hipMemcpyDeviceToHost, st));
    const double ms_enqueued = since(t_call);
    GM_HIP(hipStreamSynchronize(st));
    if (times)
        fprintf(stderr, "sssp: initialisation enqueued after %.3f ms, done after %.3f ms\n", ms_enqueued, since(t_call));
    GM_CHECK(hctrl.as<uint32_t>()[C_BAD] == 0, GM_ERR_UNSUPPORTED,
             "gm_sssp_delta_stepping: negative or NaN edge weight (the reference assumes weights >= 0)");
    if (check_weights)
        g->weights_ok.store(1, std::memory_order_relaxed);

    const double ms_setup = since(t_call);
    unsigned far_grid = gm::div_up(nwords, SSSP_BLOCK * 8);
    far_grid = far_grid > 256 ? 256 : far_grid;
    const bool use_settled = gm::measure_env("GM_SSSP_SETTLED") == nullptr || atoi(gm::measure_env("GM_SSSP_SETTLED")) != 0;
    // GM_SSSP_DONE: 0 (default) = light rounds probe every target's distance, 1 = targets taken up in earlier phases are
    // skipped by their bit once n / GM_SSSP_DONE_DIV (default 8) nodes are final, 2 = from the first phase on.  Measured at
    // scale 24 (tools/runs/r03_call39.sh, one box, bit-identical results): 9.26 ms off, 9.60 ms mode 1, 9.71 ms mode 2 — the
    // light rounds' probes of already-final targets are too few to pay for a second dependent access; off.
    const int done_mode = getenv("GM_SSSP_DONE") ? atoi(getenv("GM_SSSP_DONE")) : 0;
    const int done_div = gm::measure_env("GM_SSSP_DONE_DIV") && atoi(gm::measure_env("GM_SSSP_DONE_DIV")) > 0 ? atoi(gm::measure_env("GM_SSSP_DONE_DIV")) : 8;
    uint32_t *done_bits = done_mode ? sc->done.as<uint32_t>() : nullptr;
    const uint32_t done_min = done_mode >= 2 ? 0u : n / (uint32_t)done_div + 1u;
    // GM_SSSP_CUT=<multiples of the step> (default 8, 0: heavy rounds are whole): where the cut lies above the threshold of
    // the first phase that streamed at least m / 8 edges.  Needs the weight-ordered lists (the edges up to the cut are a slice).
    float cut_mult = order ? 8.0f : 0.0f;
    if (const char *v = getenv("GM_SSSP_CUT"))
        cut_mult = order ? (float)atof(v) : 0.0f;
    const AdvanceParams adv{adapt_lo, adapt_hi, delta / 1024.0f, 1.0e30f, cut_mult > 0.0f ? cut_mult : 0.0f,
                            (uint32_t)(g->m / 8 / 64) + 1u, order && order->in_off.p ? 1u : 0u};
    const bool stats = getenv("GM_SSSP_STATS") != nullptr;
    const int batch = stats ? 1 : 8; // rounds enqueued per host synchronisation
    auto t_prev = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < batch; ++k) {
            hipLaunchKernelGGL(sssp_round_kernel, dim3(round_grid), dim3(SSSP_BLOCK), 0, st, g->offsets, e_tgt, e_w,
                               dist.as<uint32_t>(), flags.as<uint32_t>(), wmin.as<uint32_t>(), hflags.as<uint32_t>(),
                               sc->fflags.as<uint32_t>(), settled.as<uint32_t>(), done_bits, done_min, nwords,
                               chunks.as<uint2>(), qs,
                               ctrl.as<uint32_t>(), chunk_edges, coop,
                               order && order->in_off.p ? order->in_off.as<uint32_t>() : (const uint32_t *)nullptr,
                               order && order->in_off.p ? order->in_edge.as<uint2>() : (const uint2 *)nullptr, n,
                               gm::measure_env("GM_SSSP_PULL_FILTER") && atoi(gm::measure_env("GM_SSSP_PULL_FILTER")) != 0 ? 1u : 0u);
            hipLaunchKernelGGL(sssp_chunk_kernel, dim3(grid), dim3(SSSP_BLOCK), 0, st, g->offsets, g->targets, g->weights,
                               dist.as<uint32_t>(), flags.as<uint32_t>(), wmin.as<uint32_t>(), use_settled ? settled.as<uint32_t>() : (const uint32_t *)nullptr, done_bits, done_min,
                               chunks.as<uint2>(), qs, ctrl.as<uint32_t>(), chunk_edges,
                               order ? order->targets.as<uint32_t>() : (const uint32_t *)nullptr,
                               order ? order->weights.as<float>() : (const float *)nullptr);
            hipLaunchKernelGGL(sssp_finish_kernel, dim3(far_grid), dim3(SSSP_BLOCK), 0, st, wmin.as<uint32_t>(), nwords,
                               ctrl.as<uint32_t>(), qs, adv);
        }
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpyAsync(hctrl.p, ctrl.p, C_WORDS * 4, hipMemcpyDeviceToHost, st));
        GM_HIP(hipStreamSynchronize(st));
        const uint32_t *hc = hctrl.as<uint32_t>();
        if (stats) { // batch = 1: wall clock between synchronisations is the round time
            const auto t_now = std::chrono::steady_clock::now();
            static thread_local uint32_t work_prev = 0;
            work_prev = hc[C_ROUND] <= 1 ? 0u : work_prev;
            fprintf(stderr, "sssp round %u threshold %.6f: %.3f ms, ~%u edges streamed%s\n", hc[C_ROUND],
                    __builtin_bit_cast(float, hc[C_THR]), std::chrono::duration<double, std::milli>(t_now - t_prev).count(),
                    (hc[C_WORK] - work_prev) * 64u, hc[C_HEAVY] == 2u ? " (next: far)" : hc[C_HEAVY] ? " (next: heavy)" : "");
            work_prev = hc[C_WORK];
            t_prev = t_now;
        }
        if (hc[C_DONE])
            break;
    }
    if (stats)
        fprintf(stderr, "sssp: %u rounds, %u threshold advances, width %.6f (last %.6f), ~%.1f M edge relaxations (%.2f x m)\n",
                hctrl.as<uint32_t>()[C_ROUND], hctrl.as<uint32_t>()[C_ADVANCES], width,
                __builtin_bit_cast(float, hctrl.as<uint32_t>()[C_WIDTH]),
                hctrl.as<uint32_t>()[C_WORK] * 64.0 / 1e6, g->m ? hctrl.as<uint32_t>()[C_WORK] * 64.0 / (double)g->m : 0.0);
    const double ms_rounds = since(t_call) - ms_setup;
    // distances_out: host memory (pinned memory takes the copy at link speed) or device memory
    GM_HIP(hipMemcpy(distances_out, dist.p, (size_t)n * 4, hipMemcpyDefault));
    if (times)
        fprintf(stderr, "sssp: setup %.3f ms (buffers, weight check, init), rounds %.3f ms (%u rounds), result copy %.3f ms\n",
                ms_setup, ms_rounds, hctrl.as<uint32_t>()[C_ROUND], since(t_call) - ms_setup - ms_rounds);
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

namespace gm {
void warm_sssp() // (common.hpp: the code object of this file, loaded ahead of an algorithm's first call)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&sssp_init_kernel)) != hipSuccess)
        (void)hipGetLastError();
}
} // namespace gm
