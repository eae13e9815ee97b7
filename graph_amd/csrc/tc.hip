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
// tc_dag (row id and target of every DAG entry, filled per node with wavefront help for long lists),
// tc_rows (strictly increasing lists: the bit row of L(v) in LDS, the fronts of L(u) of v's upper neighbours
// streamed against it), tc_count (everything else: a 16-lane group per DAG entry walks one prefix and
// binary-searches the other).  Integer work, HBM bound: no MFMA.
#include "common.hpp"
#include "device_utils.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

using namespace gm;

constexpr int TC_BLOCK = 256;
constexpr int TC_WAVES = TC_BLOCK / kWave;
constexpr uint32_t TC_SHORT_IDS = 65536; // ids below this are also stored in two bytes (dag16)
constexpr uint32_t TC_SHORT_PAD = 4;     // every list's 2-byte front starts on an 8-byte boundary

// Per node: |L(u)| by binary search, and a mark on the first entry of its list (row_start bitmap) so
// that the order check below can run one lane per ENTRY: checking a 600k-entry hub list with the one
// wavefront that owns the node was 24 ms of critical path at scale 24.
__global__ __launch_bounds__(TC_BLOCK) void tc_low_len_kernel(const uint32_t *__restrict__ off,
                                                              const uint32_t *__restrict__ tgt, uint32_t n,
                                                              uint32_t *__restrict__ low_len /* n+1 */,
                                                              uint32_t *__restrict__ short_len /* n+1 */,
                                                              uint32_t *__restrict__ row_start /* bit per entry */,
                                                              uint32_t *__restrict__ flags)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u <= n; u += stride) {
        if (u == n) {
            low_len[u] = 0;
            short_len[u] = 0;
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
        // how many of them are below TC_SHORT_IDS (a front of the list): those are also kept as 2-byte ids
        uint32_t lo16 = lo;
        if (u >= TC_SHORT_IDS) {
            uint32_t a = s, b = lo; // first position with tgt >= TC_SHORT_IDS
            while (a < b) {
                const uint32_t mid = a + ((b - a) >> 1);
                if (tgt[mid] < TC_SHORT_IDS)
                    a = mid + 1;
                else
                    b = mid;
            }
            lo16 = a;
        }
        short_len[u] = lo16 - s;
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

// Is the CSR its own transpose?  tc_rows_kernel finds the pairs (u, v), v < u, through v's UPPER neighbours, which is
// only right when every entry (v, u) has its mirror (u, v) — true for UndirectedCsrGraph builds, not for whatever
// sorted CSR a caller uploads or wraps.  On strictly increasing lists every ordered pair occurs at most once, so the
// lists are symmetric iff the multiset of pairs named by upper entries equals the one named by lower entries: two
// independent 64-bit fingerprints, sum over upper entries of H(u, t) minus sum over lower entries of H(t, u), both 0
// (mod 2^64).  A mismatch sends the whole count down the search path, which reads only L(u) and L(v).
__device__ __forceinline__ uint64_t tc_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(TC_BLOCK) void tc_symmetry_kernel(const uint32_t *__restrict__ off,
                                                               const uint32_t *__restrict__ tgt, uint32_t n,
                                                               unsigned long long *__restrict__ fp /* [2] */)
{
    // A wavefront takes a CHUNK of 4096 consecutive entries — consecutive lanes read consecutive entries — finds the rows of
    // the chunk's two ends by a search over the offsets, and every entry's row by a search between those two (no step at
    // all inside a hub row, a dozen steps over offsets that sit in the L1 where the rows are short).  One lane per row
    // (rounds 1-4) had every lane walk its own list, 64 cache lines per load instruction: 56 ms for the 520 M entries of RMAT
    // scale 24, more than the count itself; 64 rows per wavefront, flattened, left the degree-ordered graph's first
    // wavefront with 6 % of all entries (79 ms).
    constexpr uint32_t CHUNK = 4096;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, nwaves = gridDim.x * blockDim.x / kWave;
    const uint32_t m = off[n];
    auto row_of = [&](uint32_t i, uint32_t lo, uint32_t hi) { // the last row in [lo, hi) whose list starts at or before entry i
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (off[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        return lo;
    };
    uint64_t a = 0, b = 0;
    for (uint64_t c = wave; c * CHUNK < m; c += nwaves) {
        const uint32_t i_lo = (uint32_t)(c * CHUNK), i_hi = (uint64_t)i_lo + CHUNK < m ? i_lo + CHUNK : m;
        const uint32_t ra = row_of(i_lo, 0u, n), rb = row_of(i_hi - 1u, ra, n);
        for (uint32_t i = i_lo + lane; i < i_hi; i += kWave) {
            const uint32_t u = row_of(i, ra, rb + 1u), t = tgt[i];
            if (t != u) {
                const uint64_t key = t > u ? ((uint64_t)u << 32 | t) : ((uint64_t)t << 32 | u);
                const uint64_t h1 = tc_mix(key + 0x9E3779B97F4A7C15ull), h2 = tc_mix(key ^ 0xD6E8FEB86659FD93ull);
                a += t > u ? h1 : (uint64_t)0 - h1;
                b += t > u ? h2 : (uint64_t)0 - h2;
            }
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0 && (a | b)) {
        atomicAdd(&fp[0], (unsigned long long)a);
        atomicAdd(&fp[1], (unsigned long long)b);
    }
}

__global__ void tc_short_pad_kernel(const uint32_t *__restrict__ short_len, uint32_t n, uint32_t *__restrict__ padded)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u <= n; u += stride)
        padded[u] = u == n ? 0u : ((short_len[u] + TC_SHORT_PAD - 1u) & ~(TC_SHORT_PAD - 1u));
}

// Everything tc_rows_kernel needs of most lists in ONE 128-byte line: the header
// {first DAG entry, |L(u)|, first 2-byte entry, number of entries below TC_SHORT_IDS} and the first TC_REC_IDS
// 2-byte ids of the front (0xFFFF beyond its end).  A visit of a short list costs one random line instead of two
// (header, then the first line of the front).  Eight lanes per node, 16 bytes each.
constexpr uint32_t TC_REC_IDS = 56;
__global__ __launch_bounds__(TC_BLOCK) void tc_record_kernel(const uint32_t *__restrict__ loff,
                                                             const uint32_t *__restrict__ low_len,
                                                             const uint32_t *__restrict__ loff16,
                                                             const uint32_t *__restrict__ short_len,
                                                             const uint16_t *__restrict__ dag16, uint32_t n,
                                                             uint4 *__restrict__ rec /* 8 per node */)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (uint64_t)n * 8u; t += stride) {
        const uint32_t u = (uint32_t)(t >> 3), k = (uint32_t)(t & 7u);
        const uint32_t s16 = loff16[u], n16 = short_len[u];
        uint4 out;
        if (k == 0) {
            out = make_uint4(loff[u], low_len[u], s16, n16);
        } else {
            uint32_t w[4];
#pragma unroll
            for (uint32_t e = 0; e < 4u; ++e) {
                const uint32_t i = (k - 1u) * 8u + 2u * e;
                const uint32_t lo = i < n16 ? dag16[s16 + i] : 0xFFFFu;
                const uint32_t hi = i + 1u < n16 ? dag16[s16 + i + 1u] : 0xFFFFu;
                w[e] = lo | (hi << 16);
            }
            out = make_uint4(w[0], w[1], w[2], w[3]);
        }
        rec[t] = out;
    }
}

// dag_src[k] = u for k in [loff[u], loff[u+1]); dag_tgt[k] = the k-th lower-prefix entry; dag16 = the entries below
// TC_SHORT_IDS once more, as 2-byte ids (padding: 0xFFFF, never counted — the reader knows the length)
__global__ __launch_bounds__(TC_BLOCK) void tc_dag_kernel(const uint32_t *__restrict__ off,
                                                          const uint32_t *__restrict__ tgt,
                                                          const uint32_t *__restrict__ loff,
                                                          const uint32_t *__restrict__ loff16,
                                                          const uint32_t *__restrict__ short_len, uint32_t n,
                                                          uint32_t *__restrict__ dag_src, uint32_t *__restrict__ dag_tgt,
                                                          uint16_t *__restrict__ dag16)
{
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + kWave - 1) / kWave * kWave;
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_pad; u += stride) {
        uint32_t ls = 0, le = 0, s = 0, s16 = 0, n16 = 0, p16 = 0;
        if (u < n) {
            ls = loff[u];
            le = loff[u + 1];
            s = off[u];
            if (dag16) {
                s16 = loff16[u];
                n16 = short_len[u];
                p16 = loff16[u + 1] - s16;
            }
        }
        const uint32_t len = le - ls;
        if (len <= 32) {
            for (uint32_t i = 0; i < len; ++i) {
                const uint32_t t = tgt[s + i];
                dag_src[ls + i] = u;
                dag_tgt[ls + i] = t;
                if (i < n16)
                    dag16[s16 + i] = (uint16_t)t;
            }
            for (uint32_t i = n16; i < p16; ++i)
                dag16[s16 + i] = 0xFFFFu;
        }
        uint64_t big = __ballot(len > 32);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bu = __shfl(u, src, kWave), bls = __shfl(ls, src, kWave), ble = __shfl(le, src, kWave),
                           bs = __shfl(s, src, kWave), b16 = __shfl(s16, src, kWave), bn16 = __shfl(n16, src, kWave),
                           bp16 = __shfl(p16, src, kWave);
            for (uint32_t i = lane; i < ble - bls; i += kWave) {
                const uint32_t t = tgt[bs + i];
                dag_src[bls + i] = bu;
                dag_tgt[bls + i] = t;
                if (i < bn16)
                    dag16[b16 + i] = (uint16_t)t;
            }
            for (uint32_t i = bn16 + lane; i < bp16; i += kWave)
                dag16[b16 + i] = 0xFFFFu;
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

// ---- strictly increasing lists: the rows of the adjacency bitmap, one at a time, in LDS ---------------------------
// triangles = sum over DAG entries (u, v) of |L(u) ∩ L(v)|, and on sets |L(u) ∩ L(v)| = the number of w in L(u),
// w < v, with w in L(v).  A work item is a node v < K and a stretch of its UPPER neighbours u (the entries of N(v)
// behind its lower prefix — the undirected CSR is its own transpose): the workgroup builds the bit row of L(v) in LDS
// (v bits), then 16-lane groups take one u each and stream the front of L(u), 64 entries per step, testing every
// w < v against the row, until the list passes v.  Every probe lands in LDS and the lists are read as whole 64-byte
// lines.  (Round 1 probed a 4.3 GB triangular bitmap in HBM, one lane per DAG entry: 43 G probes at scale 24 fetched
// 761 GB in 32-byte sectors — 150 ms at 5 TB/s.)
constexpr uint32_t TCR_GROUP = 16;       // lanes per u (general path; tc_rows_kernel: template parameter)
constexpr uint32_t TCR_K_MAX = 1u << 20; // rows up to 128 KiB of LDS

// items of row v: ceil(|U(v)| / per_item) when L(v) is not empty (an empty row has no bits to hit)
__global__ void tc_item_count_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ low_len, uint32_t K,
                                     uint32_t per_item, uint32_t *__restrict__ items /* K+1 */)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v <= K; v += stride) {
        uint32_t c = 0;
        if (v < K && low_len[v]) {
            const uint32_t upper = off[v + 1] - off[v] - low_len[v];
            c = (upper + per_item - 1) / per_item;
        }
        items[v] = c;
    }
}

// item -> its row (a 19-step binary search over item_first at the head of every workgroup was ~10 us of dependent loads)
__global__ void tc_item_rows_kernel(const uint32_t *__restrict__ item_first, uint32_t K, uint32_t *__restrict__ item_row)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < K; v += stride)
        for (uint32_t i = item_first[v]; i < item_first[v + 1]; ++i)
            item_row[i] = v;
}

template <int BLOCK /* threads per work item */, uint32_t GROUP /* lanes per u */, uint32_t MLP /* loads in flight per lane */,
          uint32_t UB /* lists a group has in flight */>
__global__ __launch_bounds__(BLOCK) void tc_rows_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt,
                                                            const uint32_t *__restrict__ low_len,
                                                            const uint32_t *__restrict__ loff,
                                                            const uint4 *__restrict__ rec /* tc_record_kernel */,
                                                            const uint32_t *__restrict__ dag_tgt,
                                                            const uint16_t *__restrict__ dag16,
                                                            const uint32_t *__restrict__ item_first /* K+1, exclusive scan */,
                                                            const uint32_t *__restrict__ item_row,
                                                            uint32_t item_base /* first item of this launch */,
                                                            uint32_t n_items /* items of this launch */,
                                                            uint32_t row_words /* 4-byte words of the launch's longest row */,
                                                            uint32_t *__restrict__ ticket /* this launch's item counter */,
                                                            uint32_t per_item, uint32_t dyn,
                                                            unsigned long long *__restrict__ total)
{
    extern __shared__ uint32_t tc_row[];
    __shared__ uint64_t red[BLOCK / kWave];
    __shared__ uint32_t next_u, next_item;
    constexpr int TCR_BLOCK = BLOCK;
    // A workgroup DRAWS its items from the launch's counter and keeps its bit row between them: the row is cleared once,
    // an item sets the bits of its L(v) and takes them back afterwards.  One workgroup per item (rounds 1-4) cleared
    // v / 8 bytes of LDS for every item: 32-64 KiB for each of the 262 k one-item rows beyond 2^18, whose lists hold a
    // few dozen entries (profiles/r05_algos_profile.txt: 4.2 of the 28 ms of this kernel for those rows alone).
    for (uint32_t i = threadIdx.x; i < row_words; i += TCR_BLOCK)
        tc_row[i] = 0u;
    uint32_t count = 0;
    constexpr uint64_t GMASK = GROUP == 64 ? ~0ull : ((1ull << (GROUP & 63)) - 1ull);
    constexpr uint32_t MLP16 = MLP > 1 ? MLP / 2 : 1; // 8-byte loads of four 2-byte ids
    const uint32_t l = threadIdx.x % GROUP;
    const uint32_t gshift = (threadIdx.x & (kWave - 1)) / GROUP * GROUP; // this group's bits of a wavefront ballot
    for (;;) {
    if (threadIdx.x == 0)
        next_item = atomicAdd(ticket, 1u);
    __syncthreads(); // (also: the row is clean — the first clear, or the bits the item before took back)
    const uint32_t drawn = next_item;
    if (drawn >= n_items)
        break;
    const uint32_t item = item_base + drawn;
    const uint32_t v = item_row[item];
    const uint32_t lv = loff[v], nv = loff[v + 1] - lv;
    for (uint32_t i = threadIdx.x; i < nv; i += TCR_BLOCK) {
        const uint32_t w = dag_tgt[lv + i]; // < v: lists are strictly increasing and hold no self-loop on this path
        atomicOr(&tc_row[w >> 5], 1u << (w & 31u));
    }
    const uint32_t ubeg = off[v] + low_len[v] + (item - item_first[v]) * per_item;
    const uint32_t uend = ubeg + per_item < off[v + 1] ? ubeg + per_item : off[v + 1];
    if (threadIdx.x == 0)
        next_u = ubeg;
    __syncthreads();
    // four bits of a packed load against the row: entries base .. base + 3 of a 2-byte front of `lim` entries
    auto probe4 = [&](unsigned long long pk, uint32_t base, uint32_t lim, bool &over) {
#pragma unroll
        for (uint32_t e = 0; e < 4u; ++e) {
            const uint32_t w = (uint32_t)(pk >> (16u * e)) & 0xFFFFu;
            if (base + e < lim) {
                if (w < v)
                    count += (tc_row[w >> 5] >> (w & 31u)) & 1u;
                else
                    over = true;
            }
        }
    };
    // UB lists per group in flight: id -> list record -> rest of the list are dependent round trips, and one u at a
    // time left the kernel waiting on them (4.5 us per u and group)
    static_assert(GROUP == 8 || GROUP == 16 || GROUP == 32, "a record is 128 bytes: 16, 8 or 4 per lane");
    constexpr uint32_t RW = 32u / GROUP;  // 4-byte words of a record per lane
    constexpr uint32_t IPL = 64u / GROUP; // 2-byte positions per lane; the first 8 positions are the header
    // The groups of the workgroup draw their next UB upper neighbours from a counter in LDS: list fronts are anything
    // from empty to 1700 entries, and with a fixed assignment a workgroup (and its LDS, and its wavefront slots) stayed
    // resident until its unluckiest group was done.
    uint32_t j_static = ubeg + threadIdx.x / GROUP * UB; // dyn == 0 (measurements): a fixed share per group
    for (;;) {
        uint32_t j0 = j_static;
        if (dyn) {
            if (l == 0)
                j0 = atomicAdd(&next_u, UB);
            j0 = __shfl(j0, (int)gshift, kWave);
        }
        j_static += TCR_BLOCK / GROUP * UB;
        if (j0 >= uend)
            break;
        uint32_t uid[UB];
        uint32_t rw[UB][RW];
#pragma unroll
        for (uint32_t t = 0; t < UB; ++t) {
            const uint32_t j = j0 + t;
            uid[t] = j < uend ? tgt[j] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t t = 0; t < UB; ++t) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(rec + (size_t)uid[t] * 8u) + l * RW;
            if (uid[t] == 0xFFFFFFFFu) {
#pragma unroll
                for (uint32_t k = 0; k < RW; ++k)
                    rw[t][k] = 0u;
            } else if (RW == 4) {
                const uint4 x = *reinterpret_cast<const uint4 *>(p);
                rw[t][0] = x.x, rw[t][1] = x.y, rw[t][2 % RW] = x.z, rw[t][3 % RW] = x.w;
            } else if (RW == 2) {
                const uint2 x = *reinterpret_cast<const uint2 *>(p);
                rw[t][0] = x.x, rw[t][1 % RW] = x.y;
            } else {
                rw[t][0] = *p;
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < UB; ++t) {
            // {first DAG entry, |L(u)|, first 2-byte entry, entries below TC_SHORT_IDS}: header word k sits in lane k / RW
            uint4 mt;
            mt.x = __shfl(rw[t][0 % RW], (int)(gshift + 0u / RW), kWave);
            mt.y = __shfl(rw[t][1 % RW], (int)(gshift + 1u / RW), kWave);
            mt.z = __shfl(rw[t][2 % RW], (int)(gshift + 2u / RW), kWave);
            mt.w = __shfl(rw[t][3 % RW], (int)(gshift + 3u / RW), kWave);
            // the front of a list as 2-byte ids: half the bytes for the entries that are streamed most often (ids below
            // 65536 are 94 % of the stream at scale 24 after make_degree_ordered)
            const uint32_t lim16 = mt.w < v ? mt.w : v; // at most v entries of an increasing list are below v
            bool over = false;
#pragma unroll
            for (uint32_t e = 0; e < IPL; ++e) {
                const uint32_t pos = l * IPL + e; // 2-byte position inside the record
                const uint32_t w = (rw[t][(e / 2u) % RW] >> (16u * (e & 1u))) & 0xFFFFu;
                if (pos >= 8u && pos - 8u < lim16) {
                    if (w < v)
                        count += (tc_row[w >> 5] >> (w & 31u)) & 1u;
                    else
                        over = true;
                }
            }
            bool passed = ((__ballot(over) >> gshift) & GMASK) != 0; // an entry >= v was seen: the list is done
            uint32_t cur = TC_REC_IDS;
            bool go = !passed && cur < lim16;
            while (__ballot(go)) { // the rest of a long front, MLP16 lines per step
                unsigned long long pk[MLP16];
#pragma unroll
                for (uint32_t q = 0; q < MLP16; ++q) {
                    const uint32_t base = cur + (q * GROUP + l) * 4u;
                    pk[q] = (go && base < lim16) ? *reinterpret_cast<const unsigned long long *>(dag16 + mt.z + base) : ~0ull;
                }
                bool more_over = false;
#pragma unroll
                for (uint32_t q = 0; q < MLP16; ++q)
                    if (go)
                        probe4(pk[q], cur + (q * GROUP + l) * 4u, lim16, more_over);
                const bool grp_over = ((__ballot(more_over) >> gshift) & GMASK) != 0;
                passed = passed || grp_over;
                cur += MLP16 * GROUP * 4u;
                go = go && !grp_over && cur < lim16;
            }
            // what is left of the list: the entries from TC_SHORT_IDS up, 4-byte ids — only rows beyond TC_SHORT_IDS get here
            const uint32_t ne = mt.y < v ? mt.y : v;
            cur = mt.w;
            go = !passed && v > TC_SHORT_IDS && lim16 == mt.w && cur < ne;
            while (__ballot(go)) {
                uint32_t w[MLP];
#pragma unroll
                for (uint32_t q = 0; q < MLP; ++q) {
                    const uint32_t i = cur + q * GROUP + l;
                    w[q] = (go && i < ne) ? dag_tgt[mt.x + i] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (uint32_t q = 0; q < MLP; ++q)
                    if (w[q] < v)
                        count += (tc_row[w[q] >> 5] >> (w[q] & 31u)) & 1u;
                // ascending lists: once the last entry of the step is >= v (or past the end) the list is done
                const uint64_t past = __ballot(w[MLP - 1] >= v);
                go = go && ((past >> gshift) & GMASK) == 0;
                cur += MLP * GROUP;
            }
        }
    }
    // the item is done: its bits leave the row (every group has finished probing it)
    __syncthreads();
    const uint32_t words = (v + 31u) >> 5;
    if (nv < words) {
        for (uint32_t i = threadIdx.x; i < nv; i += TCR_BLOCK)
            tc_row[dag_tgt[lv + i] >> 5] = 0u;
    } else {
        for (uint32_t i = threadIdx.x; i < words; i += TCR_BLOCK)
            tc_row[i] = 0u;
    }
    } // items of this workgroup
    const uint64_t block_total = block_sum<uint64_t, BLOCK / kWave>((uint64_t)count, red);
    if (threadIdx.x == 0 && block_total)
        atomicAdd(total, (unsigned long long)block_total);
}

// The general path: count the entries w of L(v) that occur in L(u), by binary search.  A wavefront looks at 64 DAG
// entries at a time and hands the ones it has to do (all of them; on strictly increasing lists only those with
// v >= skip_below, the rest belong to tc_rows_kernel) to its four 16-lane groups: every lane takes one element of
// the walked prefix and searches the other prefix.  (One lane per entry walking its prefix alone was nwalk x log2
// dependent loads per entry: 50 ms for the 8 % of the scale-24 entries beyond the bitmap rows.)  On sets the
// intersection is symmetric and the shorter prefix is walked.
template <bool STRICT>
__global__ __launch_bounds__(TC_BLOCK) void tc_count_kernel(const uint32_t *__restrict__ loff,
                                                            const uint32_t *__restrict__ dag_src,
                                                            const uint32_t *__restrict__ dag_tgt, uint64_t dag_m,
                                                            uint32_t skip_below, unsigned long long *__restrict__ total)
{
    __shared__ uint64_t red[TC_WAVES];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t gi = lane / TCR_GROUP, l = lane % TCR_GROUP;
    const uint64_t dag_pad = (dag_m + kWave - 1) / kWave * kWave; // whole wavefronts iterate together
    uint64_t count = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < dag_pad; k += stride) {
        const bool live = k < dag_m;
        const uint32_t v = live ? dag_tgt[k] : 0u;
        uint64_t todo = __ballot(live && !(STRICT && v < skip_below));
        while (todo) {
            // the (gi+1)-th entry still to do is this group's
            uint64_t rest = todo;
            for (uint32_t t = 0; t < gi; ++t)
                rest &= rest - 1;
            const int src = rest ? __ffsll((unsigned long long)rest) - 1 : -1;
            for (int t = 0; t < 4 && todo; ++t)
                todo &= todo - 1;
            const uint32_t ev = __shfl(v, src < 0 ? 0 : src, kWave);
            if (src < 0)
                continue;
            const uint32_t eu = dag_src[k - lane + (uint32_t)src];
            const uint32_t *lu = dag_tgt + loff[eu];
            const uint32_t *lv = dag_tgt + loff[ev];
            const uint32_t nu = loff[eu + 1] - loff[eu], nv = loff[ev + 1] - loff[ev];
            const uint32_t *walk = lv, *probe = lu;
            uint32_t nwalk = nv, nprobe = nu;
            if (STRICT && nu < nv) {
                walk = lu;
                probe = lv;
                nwalk = nu;
                nprobe = nv;
            }
            for (uint32_t i = l; i < nwalk; i += TCR_GROUP)
                count += tc_contains(probe, nprobe, walk[i]) ? 1u : 0u;
        }
    }
    const uint64_t block_total = block_sum<uint64_t, TC_WAVES>(count, red);
    if (threadIdx.x == 0 && block_total)
        atomicAdd(total, (unsigned long long)block_total);
}

} // namespace

// What a triangle count derives from the graph alone: the DAG of lower prefixes and, for strictly increasing lists,
// its 2-byte fronts and list records.  Immutable once built; kept in the CSR handle (gm_csr::tc_dag) so that a second
// count on the same graph — the reference's app times global_triangle_count in a loop — skips 5 ms of construction
// and a dozen hipMalloc / hipFree pairs at scale 24.  GM_TC_NOCACHE=1 builds a private one (measurements).
struct gm::TcDag {
    gm::DevBuf low_len, loff, dag_src, dag_tgt, dag16, rec;
    uint32_t flags = 0; // 2: some list has equal neighbours or a self-loop (general path only)
    uint32_t dag_m = 0;
    bool rows_ok = false; // dag16 / rec exist: tc_rows_kernel can run
    bool symmetric = true; // strictly increasing lists only: every entry has its mirror (else: search path only)
};

namespace {

int tc_prepare(const gm_csr *g, gm::TcDag &d)
{
    const uint32_t n = (uint32_t)g->n;
    gm::DevBuf &low_len = d.low_len, &loff = d.loff;
    gm::DevBuf short_len, loff16, ctrl;
    GM_TRY(low_len.alloc(((size_t)n + 1) * 4));
    GM_TRY(loff.alloc(((size_t)n + 1) * 4));
    GM_TRY(short_len.alloc(((size_t)n + 1) * 4));
    GM_TRY(loff16.alloc(((size_t)n + 1) * 4));
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
                           low_len.as<uint32_t>(), short_len.as<uint32_t>(), row_start.as<uint32_t>(),
                           ctrl.as<uint32_t>() + 2);
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
        gm::DevBuf tmp, padded;
        GM_TRY(tmp.alloc(tmp_bytes));
        GM_TRY(padded.alloc(((size_t)n + 1) * 4));
        GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, low_len.as<uint32_t>(), loff.as<uint32_t>(), 0u,
                                       (size_t)n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        hipLaunchKernelGGL(tc_short_pad_kernel, dim3(grid), dim3(TC_BLOCK), 0, 0, short_len.as<uint32_t>(), n,
                           padded.as<uint32_t>());
        GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, padded.as<uint32_t>(), loff16.as<uint32_t>(), 0u,
                                       (size_t)n + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        GM_HIP(hipDeviceSynchronize());
    }
    uint32_t short_m = 0;
    GM_HIP(hipMemcpy(&d.flags, ctrl.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost));
    GM_CHECK((d.flags & 1u) == 0, GM_ERR_UNSUPPORTED,
             "gm_triangle_count: neighbour lists are not sorted (use CsrLayout::Sorted or Deduplicated)");
    GM_HIP(hipMemcpy(&d.dag_m, loff.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
    if (d.dag_m == 0)
        return GM_OK;
    // the 2-byte fronts and the records are only read by tc_rows_kernel (strictly increasing lists)
    d.rows_ok = (d.flags & 2u) == 0; // they need 2 B x short entries + 128 B x nodes beside the DAG
    if (d.rows_ok) { // ... and a CSR that is its own transpose (tc_symmetry_kernel)
        gm::DevBuf fp;
        GM_TRY(fp.alloc(16));
        GM_HIP(hipMemset(fp.p, 0, 16));
        hipLaunchKernelGGL(tc_symmetry_kernel, dim3(grid), dim3(TC_BLOCK), 0, 0, g->offsets, g->targets, n,
                           fp.as<unsigned long long>());
        GM_HIP(hipGetLastError());
        unsigned long long h[2] = {0, 0};
        GM_HIP(hipMemcpy(h, fp.p, 16, hipMemcpyDeviceToHost));
        d.symmetric = (h[0] | h[1]) == 0;
        d.rows_ok = d.symmetric;
    }
    GM_HIP(hipMemcpy(&short_m, loff16.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
    GM_TRY(d.dag_src.alloc((size_t)d.dag_m * 4));
    GM_TRY(d.dag_tgt.alloc((size_t)d.dag_m * 4));
    if (d.rows_ok && (d.dag16.alloc((size_t)short_m * 2 + 16) != GM_OK || d.rec.alloc((size_t)n * 8 * sizeof(uint4)) != GM_OK)) {
        // an accelerator, not a requirement: when HBM is short the whole count takes the search path
        (void)hipGetLastError();
        d.dag16.release();
        d.rec.release();
        d.rows_ok = false;
    }
    hipLaunchKernelGGL(tc_dag_kernel, dim3(grid), dim3(TC_BLOCK), 0, 0, g->offsets, g->targets, loff.as<uint32_t>(),
                       loff16.as<uint32_t>(), short_len.as<uint32_t>(), n, d.dag_src.as<uint32_t>(), d.dag_tgt.as<uint32_t>(),
                       d.rows_ok ? d.dag16.as<uint16_t>() : (uint16_t *)nullptr);
    if (d.rows_ok) {
        unsigned rgrid = gm::div_up((uint64_t)n * 8, TC_BLOCK);
        rgrid = rgrid > 256 * 64 ? 256 * 64 : rgrid;
        hipLaunchKernelGGL(tc_record_kernel, dim3(rgrid), dim3(TC_BLOCK), 0, 0, loff.as<uint32_t>(), low_len.as<uint32_t>(),
                           loff16.as<uint32_t>(), short_len.as<uint32_t>(), d.dag16.as<uint16_t>(), n, d.rec.as<uint4>());
    }
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize()); // short_len / loff16 are released on return
    return GM_OK;
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
    std::shared_ptr<const gm::TcDag> dag;
    const bool cache = getenv("GM_TC_NOCACHE") == nullptr || atoi(getenv("GM_TC_NOCACHE")) == 0;
    if (cache) {
        std::lock_guard<std::mutex> lock(g->cache_mu);
        dag = g->tc_dag;
    }
    if (!dag) {
        auto fresh = std::make_shared<gm::TcDag>();
        GM_TRY(tc_prepare(g, *fresh));
        dag = fresh;
        if (cache) {
            std::lock_guard<std::mutex> lock(g->cache_mu);
            if (!g->tc_dag)
                g->tc_dag = dag;
        }
    }
    const uint32_t flags = dag->flags, dag_m = dag->dag_m;
    const bool rows_ok = dag->rows_ok;
    if (dag_m == 0)
        return GM_OK;
    const gm::DevBuf &low_len = dag->low_len, &loff = dag->loff, &dag_src = dag->dag_src, &dag_tgt = dag->dag_tgt,
                     &dag16 = dag->dag16, &rec = dag->rec;
    gm::DevBuf ctrl; // the count (8 bytes, 16 reserved), then one item counter per tc_rows_kernel launch
    GM_TRY(ctrl.alloc(16 + 64 * 4));
    GM_HIP(hipMemset(ctrl.p, 0, 16 + 64 * 4));
    unsigned cgrid = gm::div_up(dag_m, TC_BLOCK);
    if (cgrid > 256 * 16)
        cgrid = 256 * 16;
    unsigned long long *d_total = reinterpret_cast<unsigned long long *>(ctrl.p);
    if (flags & 2u) {
        hipLaunchKernelGGL(tc_count_kernel<false>, dim3(cgrid), dim3(TC_BLOCK), 0, 0, loff.as<uint32_t>(),
                           dag_src.as<uint32_t>(), dag_tgt.as<uint32_t>(), (uint64_t)dag_m, 0u, d_total);
    } else {
        // strictly increasing lists: rows v < K by tc_rows_kernel (bit row of L(v) in LDS), the rest by search.
        // GM_TC_K overrides K (0: search only), GM_TC_ITEM the upper neighbours per work item.
        // K = 2^19: 64 KiB rows, two workgroups = 32 wavefronts per CU (2^20 halves the occupancy: 100 vs 54 ms at scale 24)
        uint32_t K = n < (1u << 19) ? n : (1u << 19);
        if (const char *e = getenv("GM_TC_K")) {
            const uint32_t cap = n < TCR_K_MAX ? n : TCR_K_MAX;
            K = (uint32_t)atoll(e) < cap ? (uint32_t)atoll(e) : cap;
        }
        if (!rows_ok)
            K = 0;
        uint32_t per_item = 2048;
        if (const char *e = getenv("GM_TC_ITEM"))
            if (atoll(e) >= 64)
                per_item = (uint32_t)atoll(e);
        gm::DevBuf items, item_first, item_row;
        std::vector<uint32_t> first_host;
        if (K) {
            GM_TRY(items.alloc(((size_t)K + 1) * 4));
            GM_TRY(item_first.alloc(((size_t)K + 1) * 4));
            hipLaunchKernelGGL(tc_item_count_kernel, dim3(gm::div_up((uint64_t)K + 1, 256)), dim3(256), 0, 0, g->offsets,
                               low_len.as<uint32_t>(), K, per_item, items.as<uint32_t>());
            size_t tmp_bytes = 0;
            GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, items.as<uint32_t>(), item_first.as<uint32_t>(), 0u,
                                           (size_t)K + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
            gm::DevBuf tmp;
            GM_TRY(tmp.alloc(tmp_bytes));
            GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, items.as<uint32_t>(), item_first.as<uint32_t>(), 0u,
                                           (size_t)K + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
            first_host.resize((size_t)K + 1);
            GM_HIP(hipMemcpy(first_host.data(), item_first.p, ((size_t)K + 1) * 4, hipMemcpyDeviceToHost));
            GM_TRY(item_row.alloc(((size_t)first_host[K] + 1) * 4));
            hipLaunchKernelGGL(tc_item_rows_kernel, dim3(gm::div_up(K, 256)), dim3(256), 0, 0, item_first.as<uint32_t>(), K,
                               item_row.as<uint32_t>());
        }
        // One launch per power-of-two range of rows, with the LDS its longest row needs: the hub rows (short rows,
        // most of the work) are not held to the occupancy of the 64 KiB rows at the far end.
        // GM_TC_SHAPE="<threads per item>,<lanes per u>,<loads in flight>" picks another instantiation (measurements)
        const uint32_t dyn = getenv("GM_TC_DYN") ? (uint32_t)atoi(getenv("GM_TC_DYN")) : 1u;
        // measured best at scale 24 (1024 x 16: +12 %, 256 x 16: +11 %; four lists per draw: 45.9 ms against 39.3 for one —
        // with the lists drawn from the counter the balance is worth more than the overlap of the dependent loads)
        int shape_b = 512, shape_g = 8, shape_m = 4, shape_u = 1;
        // (2, 4 or 8 lists per draw for the hub rows' launches changed nothing, 39.7-40.7 against 40.1 ms: round 2, removed)
        if (const char *e = getenv("GM_TC_SHAPE"))
            (void)sscanf(e, "%d,%d,%d,%d", &shape_b, &shape_g, &shape_m, &shape_u);
        // workgroups of a launch: as many as the chip holds at once (LDS and wavefront slots), times GM_TC_WAVES (default 2:
        // a workgroup that arrives late finds the counter run out and leaves) — never more than the items
        const uint32_t tc_waves = gm::measure_env("GM_TC_WAVES") && atoi(gm::measure_env("GM_TC_WAVES")) > 0 ? (uint32_t)atoi(gm::measure_env("GM_TC_WAVES")) : 2u;
        int n_cus = 256;
        (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, g->device);
        auto wgs_of = [&](uint32_t n_items, size_t lds, int block) {
            const uint32_t by_lds = (uint32_t)((160u * 1024u) / (lds + 1024u)), by_waves = 2048u / (uint32_t)block;
            const uint32_t per_cu = std::max(1u, std::min(by_lds, by_waves));
            const uint64_t wgs = (uint64_t)n_cus * per_cu * tc_waves;
            return (uint32_t)std::min<uint64_t>(n_items, gm::measure_env("GM_TC_PERSIST") && atoi(gm::measure_env("GM_TC_PERSIST")) == 0 ? n_items : wgs);
        };
        uint32_t launch_no = 0; // (at most 1 + log2(K / 16384) <= 7 launches)
        // GM_TC_STREAMS=1: every range on a stream of its own (ordered behind the set-up work on the null stream), so
        // that the tail of one range — a few workgroups on their last, long items — runs under the next range's start
        static hipStream_t pool[16][8] = {};
        static std::mutex pool_mu;
        const bool multi = gm::measure_env("GM_TC_STREAMS") && atoi(gm::measure_env("GM_TC_STREAMS")) != 0; // measured (tools/runs/r05_call04.sh): 30.1 against 29.0 ms at scale 24 — the 64 KiB rows take CUs from the hub rows: off
        auto stream_of = [&](uint32_t k) -> hipStream_t {
            if (!multi || g->device < 0 || g->device >= 16)
                return (hipStream_t)0;
            std::lock_guard<std::mutex> lock(pool_mu);
            hipStream_t &st = pool[g->device][k % 8u];
            if (!st && hipStreamCreate(&st) != hipSuccess) { // (blocking streams: they wait for the null stream's earlier work)
                (void)hipGetLastError();
                st = nullptr;
            }
            return st;
        };
        for (uint32_t v_lo = 0; v_lo < K;) {
            uint32_t v_hi = v_lo < (1u << 14) ? (1u << 14) : v_lo * 2u;
            v_hi = v_hi < K ? v_hi : K;
            const uint32_t n_items = first_host[v_hi] - first_host[v_lo];
            const size_t lds = (size_t)((v_hi + 31u) / 32u) * 4;
            if (n_items) {
#define GM_TC_ROWS(B_, G_, M_, U_)                                                                                      \
    do {                                                                                                                \
        GM_CHECK(launch_no < 64, GM_ERR_INVALID, "gm_triangle_count: %u launches over the rows (ctrl holds 64 item counters)", launch_no); \
        GM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tc_rows_kernel<B_, G_, M_, U_>),                     \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TCR_K_MAX / 8));                         \
        hipLaunchKernelGGL((tc_rows_kernel<B_, G_, M_, U_>), dim3(wgs_of(n_items, lds, B_)), dim3(B_), lds, stream_of(launch_no), g->offsets, \
                           g->targets, low_len.as<uint32_t>(), loff.as<uint32_t>(), rec.as<uint4>(),                   \
                           dag_tgt.as<uint32_t>(), dag16.as<uint16_t>(), item_first.as<uint32_t>(),                    \
                           item_row.as<uint32_t>(), first_host[v_lo], n_items, (uint32_t)(lds / 4),                    \
                           reinterpret_cast<uint32_t *>(ctrl.p) + 4 + launch_no, per_item, dyn, d_total);              \
        ++launch_no;                                                                                                    \
    } while (0)
                // the product shape, and the ones tests/test_gpu_parity.py walks through ("every way the work can be split
                // gives the same count"); anything else runs as the product shape
                if (shape_b == 1024 && shape_g == 16)
                    GM_TC_ROWS(1024, 16, 4, 4);
                else if (shape_b == 1024 && shape_g == 8)
                    GM_TC_ROWS(1024, 8, 4, 1);
                else if (shape_b == 512 && shape_g == 16)
                    GM_TC_ROWS(512, 16, 4, 2);
                else if (shape_b == 512 && shape_g == 8 && shape_m == 8)
                    GM_TC_ROWS(512, 8, 8, 8);
                else if (shape_b == 512 && shape_g == 8 && shape_u == 4)
                    GM_TC_ROWS(512, 8, 4, 4);
                else if (shape_b == 256 && shape_g == 16)
                    GM_TC_ROWS(256, 16, 4, 4);
                else if (shape_b == 256 && shape_g == 8)
                    GM_TC_ROWS(256, 8, 4, 4);
                else if (shape_b == 128 && shape_g == 8)
                    GM_TC_ROWS(128, 8, 4, 4);
                else
                    GM_TC_ROWS(512, 8, 4, 1);
#undef GM_TC_ROWS
            }
            v_lo = v_hi;
        }
        hipLaunchKernelGGL(tc_count_kernel<true>, dim3(cgrid), dim3(TC_BLOCK), 0, stream_of(7), loff.as<uint32_t>(),
                           dag_src.as<uint32_t>(), dag_tgt.as<uint32_t>(), (uint64_t)dag_m, K, d_total);
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize()); // the item tables are released on scope exit
    }
    GM_HIP(hipGetLastError());
    unsigned long long total = 0;
    GM_HIP(hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost));
    *triangles_out = total;
    return GM_OK;
}

namespace gm {
void warm_tc() // (common.hpp: the code object of this file, loaded ahead of an algorithm's first call)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&tc_order_kernel)) != hipSuccess)
        (void)hipGetLastError();
}
} // namespace gm
