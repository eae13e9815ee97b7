// pagerank_pb.hip — PageRank sweeps by propagation blocking (gfx950 / wave64).
//
// Why: the pull sweep's gather out_scores[v] (crates/algos/src/page_rank.rs:143-146) touches one
// 128-byte line per 4 useful bytes once the vector outgrows the caches; measured on MI355X
// (tools/membench.hip) a random 4-byte gather runs at ~55 G/s from HBM and ~200 G/s even when the
// table sits in L2 — the sweep is bound by line transactions, not by bytes.  Only LDS serves
// random 4-byte accesses fast enough, so the sweep is restructured so that every random access
// lands in LDS and everything that touches HBM is a sequential stream:
//
//   pb_bin_kernel    one workgroup per chunk (24576 entries) of a SOURCE TILE (2^s_log = 16384 or 32768
//                    consecutive ids of x): the tile's out_scores are loaded into LDS (64 / 128 KiB,
//                    coalesced); the tile's edges — stored once, at plan creation, as 2-byte local source
//                    ids grouped by destination bin — are streamed and each edge's value xs[src] is
//                    appended to its bin's slice of the `vals` stream (float4 writes, runs of (tile, bin)
//                    segments).  Software-pipelined: ids of the next step are requested before the stores
//                    of this one.
//   pb_accum_kernel  one workgroup per DESTINATION BIN (R <= 16384 consecutive rows; over-long bins are
//                    sliced): streams the bin's values and 2-byte accumulator slots and accumulates into
//                    an LDS array of 64-bit fixed-point sums (ds_add_u64, scale 2^62: exact,
//                    order-independent); edges of the H most frequent ("hot") sources skip the value
//                    stream — 4-byte (slot, hot index) records read against an LDS table of their
//                    out_scores; then the fused epilogue of the reference (new score, |delta|, out_score).
//   pb_err_kernel    sums the per-bin f64 errors in index order.
//
// The row sum is therefore the exactly rounded sum of the f32 out_scores: deterministic, identical
// for any partition of the rows over GPUs, and closer to the real-number fixed point than any f32
// summation order.
//
// HUB ROWS.  The reference adds a row's in-neighbours left to right in f32 (page_rank.rs:143-146), and on a
// long row that order has a SYSTEMATIC drift: thousands of in-neighbours carry the same out_score
// ((1-d)/n / out_degree of nodes without in-edges), every one of them is rounded the same way against the
// running sum, and the errors add up instead of cancelling (measured at RMAT scale 26: the reference's own
// score of the 854,315-in-edge row is 8.5e-4 away from the exact row sum, and every node that row points to
// inherits that).  Matching the reference within 1e-5 therefore means reproducing its rounding, not being
// more exact.  Rows with at least `hub_deg` in-edges (default 4096, GM_PB_HUB_DEG) are taken out of the
// ordinary bins: consecutive hub rows form HUB GROUPS (<= 64 rows, about one ordinary bin's worth of terms),
// every group is one more "bin" of the value stream (so (tile, group) segments are as long as (tile, bin)
// ones), and no hub row uses the hot path — all its terms arrive in the stream in ascending source order, the
// CSR order of the Sorted / Deduplicated layouts.  Their sums are then COMPUTED the reference's way, bit for bit
// for the same out_scores (rounds 2 and 3 imitated the order with integer counts of ulps per 4096-entry step,
// within ~3e-6 and at the price of as many vector instructions as the accumulate kernel itself):
//   pb_hubseq_kernel   rows below `hub_long` terms (8192 or more, see pb_build): a group's 2048-entry blocks are made row-major in LDS by a
//                      permutation fixed at plan time, and lane g of one wavefront adds row g's terms in order,
//                      one v_add_f32 per term;
//   pb_hublong_kernel  longer rows, one workgroup each: inside one binade of the running sum S = J ulp, adding a term
//                      v moves J by rint(v / ulp) — up or down on a tie as the parity of J says — so a run of terms is
//                      a function (count from even J, count from odd J) and runs compose associatively: 512 threads
//                      take 16 terms each, one scan gives every thread its J, and the thread in whose run S leaves
//                      the binade adds its 16 terms the slow way before the rest is redone on the coarser grid.
//
// HBM traffic per edge and sweep: cold 2 B (source id) + 4 B (value write) + 4 B (value read) + 2 B (slot)
// = 12 B, hot 4 B, all streaming, against 8 B "algorithmic" of which 4 B are a random gather.
#include "pagerank.hpp"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace gm {

namespace {

constexpr int PB_S_LOG_MAX = 15;     // sources per tile = 2^s_log, s_log 14 (x tile = 64 KiB of LDS) or 15 (128 KiB)
constexpr int PB_BIN_BLOCK = 1024;
constexpr int PB_ACC_BLOCK = 1024;
constexpr uint32_t PB_VEC = 4;                  // segments are padded to multiples of 4 entries in both streams
constexpr uint32_t PB_WBLK = kWave * PB_VEC;    // entries one wavefront covers per step (256)
constexpr uint16_t PB_NULL = 0xFFFFu;
constexpr size_t PB_ACC_STATIC = 512;  // static LDS of pb_accum_kernel, rounded up
constexpr uint32_t PB_HUB_MAX = 64;  // rows of a hub group (one lane of a wavefront each)
constexpr uint16_t PB_HUBROW = 0xFFFEu; // cidx of a hub row: its sum is produced by pb_hubseq_kernel / pb_hublong_kernel, not by its bin
constexpr uint16_t PB_FLAG = 0x8000u;
constexpr uint32_t PB_SEQ_WG = 256;  // threads of a pb_hubseq_kernel workgroup (wavefront 0 walks, all four stage) ...
constexpr uint32_t PB_SEQ_WG_WIDE = 512; // ... and of the launches that are a sweep's critical path (a part's hub rows, a slice's): see the kernel
constexpr uint32_t PB_SEQ_PAD = 16;  // a row's stretch of the staged block is padded to 16 floats (4 x ds_read_b128 per step)
constexpr uint32_t PB_SEQ_STEP = 2048; // entries of a block of pb_hubseq_kernel: what one round of loads covers
constexpr uint32_t PB_SEQ_CAP = 2040;  // terms of a block (its stretch of the value stream may start 3 entries into a float4)
constexpr uint32_t PB_SEQ_HOT = 1024;  // hot records per block that travel through the prefetch registers (4 / 2 per thread)
constexpr uint32_t PB_SEQ_BUF = PB_SEQ_STEP + PB_HUB_MAX * (PB_SEQ_PAD - 1); // floats: 2048 terms + the rows' padding (12 KiB)
// LDS left free beside an accumulate workgroup: ONE pb_hubseq_kernel workgroup (12.1 KiB) AND one pb_hublong_kernel workgroup
// (9.3 KiB).  With room for only one of the two (20 KiB blocks / an 18 KiB turning buffer, the first version) the long rows
// waited for the lane walks to leave the CUs: at scale 22 the hub phase was 67 + 55 us instead of max(67, 55).
constexpr size_t PB_HUB_ROOM = 22016;
constexpr uint32_t PB_LONG_WG = 512;  // threads of a pb_hublong_kernel workgroup
constexpr uint32_t PB_LONG_PER = 16;  // consecutive terms per thread and pass (32: 174 VGPRs — no room beside the accumulate wavefronts)
constexpr int PB_TIERS_DEFAULT = 16;  // at most this many tiers of hot sources unless GM_PB_TIERS says otherwise
constexpr float PB_FIX_SCALE = 4611686018427387904.0f;     // 2^62
constexpr float PB_FIX_INV = 2.168404344971008868e-19f;    // 2^-62

} // namespace

// the sweep's error summed by the last workgroup of the accumulate / hub launches (pb_err_fold below)
struct PbErrFold {
    uint32_t *ctr = nullptr;       // null: off
    uint32_t total = 0;            // workgroups that take part: the accumulate launch's + the hub launches'
    uint32_t count = 0;            // error slots: the bins', then the hub groups' / rows'
    const double *slots = nullptr;
    double *out = nullptr;
};

// mutable per-engine buffers: engines on one shared plan never touch each other's data
struct PbScratch {
    // hub rows summed WITH THE PART THEIR ROWS LIE IN (pb_set_parts(.., hub_by_part): block-Gauss-Seidel sweeps) instead of all
    // of them with part 0: per part the lane-walk groups (indices into hub_items behind the long ones) and the long rows'
    // items (indices into long_items, every row's in pass order), and an item counter per part
    bool hub_by_part = false;
    DevBuf part_seq_list, part_long_list, part_long_tickets;
    std::vector<uint32_t> part_seq_off, part_long_off;
    DevBuf fold_ctr;  // u32: tickets drawn by the workgroups of a whole sweep's accumulate and hub launches (self-resetting)
    PbErrFold fold;   // what the launches being enqueued right now are given (pb_sweep_main sets and clears it)
    // row parts of a partitioned sweep (gm_pr_set_parts): the items of part k are
    // part_items[part_off[k] .. part_off[k+1]), each part longest-first
    DevBuf part_items;
    std::vector<uint32_t> part_off;
    // regions of the x vector given as lists of tile ranges (gm_pr_set_bin_regions): the phase-1 work items of region r
    // are region_items[region_off[r] .. region_off[r + 1])
    DevBuf region_items;
    std::vector<uint32_t> region_off;
    // how the value stream got its memory (diagnostics, gm_pr_plan_info): bin-kernel time of the fastest / slowest timed
    // draw in us, draws timed, 64 MiB pieces the arena was grown by for it
    uint32_t draw_best_us = 0, draw_worst_us = 0, draws_timed = 0, grown_pieces = 0;
    DevBuf vals_raw; // backing allocation of the value stream
    std::shared_ptr<DevBuf> vals_shared; // GM_PB_VALS_SHARE (measurement): one allocation behind the streams of several engines
    float *vals = nullptr; // f32[Mv] per-edge values, bin-major, segments padded to 4
    DevBuf partials; // u64[slots x R] partial LDS accumulators of split bins
    DevBuf tickets;  // u32[B]    arrival counters of split bins (self-resetting)
    DevBuf bin_err;  // f64[B + G]
    hipStream_t side = nullptr;           // the hub groups run beside the ordinary bins
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t chain = nullptr;          // ... and the long rows (pb_hublong_kernel) beside the other hub groups
    hipEvent_t ev_chain_fork = nullptr, ev_chain_join = nullptr;
    ~PbScratch()
    {
        if (chain)
            (void)hipStreamDestroy(chain);
        if (ev_chain_fork)
            (void)hipEventDestroy(ev_chain_fork);
        if (ev_chain_join)
            (void)hipEventDestroy(ev_chain_join);
        if (vals_shared)
            vals_raw.p = nullptr, vals_raw.bytes = 0;
        if (side)
            (void)hipStreamDestroy(side);
        if (ev_fork)
            (void)hipEventDestroy(ev_fork);
        if (ev_join)
            (void)hipEventDestroy(ev_join);
    }
    DevBuf hot_x;    // f32[H]    out_scores of the hot sources, refreshed every sweep
    // pb_hublong_kernel, a row over several workgroups: the item counter (u64), one hand-off word per item, the sums at the
    // pass boundaries of the sweep before (what the next sweep predicts from), the launch number
    DevBuf long_state; // u64 ticket | u64 handoff[n_long_items] | f32 sbs[long_sbs_len]
    uint32_t long_epoch = 0;
};

struct PbItem {
    uint32_t bin, q0, q1;          // value-stream range [q0, q1) of bin `bin`
    uint32_t h0, h1;               // hot-edge range of this item
    uint32_t nparts, slot0, part;  // slices of this bin, first partial-accumulator slot of the bin, this slice
};

struct PbHubItem {
    uint32_t q0, q1; // value-stream range of the group (virtual bin B + group)
    uint32_t nh;     // rows of the group: slots [0, nh)
    uint32_t row0;   // their ids: hub_rows[row0 .. row0 + nh)
    uint32_t group;
};

// an item of pb_hublong_kernel (described there)
struct PbLongItem {
    uint32_t row;    // the row's entry of hub_items
    uint32_t pass0;  // first pass (super-block of PB_LONG_WG x PB_LONG_PER stream entries) of the row this item covers
    uint32_t npass;  // ... and how many (<= PB_LONG_PMAX)
    uint32_t prev;   // the item whose S this one starts from; 0xFFFFFFFF: the row's first item (S = 0)
    uint32_t sb0;    // the row's first entry of the per-engine array of pass-boundary sums (its passes + 1 entries)
    uint32_t flags;  // 1: the row's last item (finishes the row); 2: the row has other items (pairs are formed ahead)
};

struct PbPlan {
    uint32_t n_local = 0, m = 0;
    uint64_t x_len = 0;
    int rb = 0;            // log2(rows per bin)
    int s_log = 14;        // log2(sources per tile)
    uint32_t R = 0, B = 0; // rows per bin, bins
    uint32_t Racc = 0;     // accumulators per bin = max number of rows with in-edges in one bin (<= R)
    DevBuf cidx;           // u16[n]  accumulator slot of each row inside its bin, PB_NULL = no in-edges
    uint32_t hub_deg = 0;  // rows with >= hub_deg in-edges are summed in the reference's order (0 = feature off)
    uint32_t n_hub = 0;    // such rows
    uint32_t G = 0;        // hub groups: virtual bins B .. B + G - 1 of the streams
    uint64_t hub_edges = 0;
    DevBuf hub_rows;       // u32[n_hub] row id of every hub row: the walked ones ascending, then the long ones ascending (pb_build)
    DevBuf hub_first;      // u32[G+1]   first hub row (index into hub_rows) of every group
    DevBuf hub_items;      // PbHubItem[G]: the G_long groups that are one long row first, each part longest first
    uint32_t hub_long = 8192;  // rows with at least this many in-edges are a group of their own (pb_hublong_kernel): see pb_build
    uint32_t G_long = 0;
    uint64_t long_terms = 0;   // in-edges of the long rows
    // hub rows whose lists are not ascending (CsrLayout::Unsorted): summed in CSR order through `hub_gidx` (pb_hublong_kernel<true>)
    uint32_t hub_csr = 0;
    uint32_t err_slots = 0;    // error sums behind the B bins': one per hub group, or (hub_csr) one per hub row
    DevBuf hub_gidx;           // u32[..] per hub row (4-aligned stretches): value-stream position of its k-th in-neighbour's out_score
    DevBuf csr_items;          // PbHubItem[n_hub]: {first, end of the row's stretch of hub_gidx, 1, index in hub_rows, error slot}
    std::vector<uint32_t> hub_degs_host;
    DevBuf long_rows;          // PbHubItem[n_long_rows]: the rows pb_hublong_kernel sums ({q0, q1: the group's stretch; nh: the row's SLOT; row0; error slot})
    uint32_t n_long_rows = 0;
    DevBuf long_items;         // PbLongItem[n_long_items]: the long rows cut into items of a few passes, row by row (longest row first)
    uint32_t n_long_items = 0;
    uint32_t long_sbs_len = 0; // entries of an engine's array of pass-boundary sums (every long row: its passes + 1)
    std::vector<uint32_t> hub_first_host;
    std::vector<uint8_t> hub_long_host; // per group: 1 = one long row
    std::vector<PbHubItem> hub_items_host;
    std::vector<PbHubItem> long_rows_host;   // host copy of long_rows
    std::vector<PbLongItem> long_items_host; // ... and of long_items (declared below)
    // the other hub groups, hub_items[G_long .. G): walked by pb_hubseq_kernel with one lane per row.  Their part of p2_dst
    // holds, instead of the row slot, the entry's place in the row-major LDS arrangement of its 2048-entry block
    // (pb_hubseq_layout_kernel).
    uint32_t seq_blocks = 0;
    DevBuf seq_blk_first;  // u32[G - G_long + 1] first block of each such group, in hub_items order
    DevBuf seq_rows;       // u32[seq_blocks x 64] per block and row: first LDS slot << 16 | terms of the row in this block
    // ... and their terms from HOT sources do not pass the value stream at all (12 B per edge): a block is PB_SEQ_CAP consecutive
    // entries of the group's terms in source order, cold ones (a stretch of the value stream, padding included) and hot ones
    // (4-byte records: place << 18 | hot rank, the value gathered from hot_x) merged
    DevBuf seq_blk;        // uint4[seq_blocks]: {first, end of the block's stretch of the value stream, first, end of its hot records}
    DevBuf hh_ent;         // u32[Mhh] hot records of the hub groups, block-major
    uint64_t Mhh = 0;
    double build_ms = 0.0; // wall time of pb_build (device work included)
    uint32_t NT = 0;       // source tiles
    uint32_t NS = 0;       // non-empty (tile, bin) segments
    uint64_t Mp = 0;       // padded length of the phase-1 stream
    uint64_t Mv = 0;       // padded length of the value stream
    uint32_t NW = 0;       // phase-1 workgroups
    uint32_t chunk = 0;    // phase-1 entries per workgroup (multiple of 256)
    int device = 0;
    DevBuf p1_src;      // u16[Mp]   local source id | PB_FLAG on the first entry of a segment; padding entries are
                        //           unflagged ids: they belong to the segment before them and land in its padding slots
    DevBuf chunk_seg;   // u32[Mp/256] segments started before each 256-entry wavefront block
    DevBuf delta;       // u32[NS]   slot = p + delta[segment]   (mod 2^32, a multiple of 4)
    DevBuf tile_p;      // u32[NT+1] phase-1 range of each tile (multiples of 256)
    DevBuf wg_tile;     // u32[NW]   tile of each phase-1 workgroup
    DevBuf wg_p0;       // u32[NW]   first phase-1 entry of each workgroup
    DevBuf wg_tile_g, wg_p0_g; // GM_PB_WG_GROUP=G (measurement): the same items in (tile / G, chunk, tile % G) order, whole sweeps only
    DevBuf p2_dst;      // u16[Mv]   local row id inside the bin, PB_NULL = padding
    DevBuf bin_v;       // u32[B+1]  value range of each bin (multiples of 4)
    DevBuf items;       // PbItem[NI] accumulate work items (a bin, or a slice of an over-long bin), longest first
    uint32_t slots = 0; // partial-accumulator slots needed by split bins
    uint32_t NI = 0;    // accumulate workgroups
    // hot sources: the H most frequent sources of this rank's edges skip the value stream; their
    // out_scores are staged in LDS by the accumulate kernel and gathered there
    // ... in TIERS: the LDS beside the accumulators holds the out_scores of H sources at a time, and an accumulate
    // workgroup walks through T such tables one after the other (tier t = the sources ranked [t H, (t + 1) H) by
    // frequency), each followed by the bin's hot edges of that tier.  A further tier costs every workgroup one table
    // load from L2 (4 H bytes) and two barriers, and moves its edges from 12 to 4 streamed bytes.
    uint32_t H = 0;     // hot sources per tier (0 = feature off)
    uint32_t T = 1;     // tiers
    uint32_t Htot = 0;  // hot sources in all (<= T * H; the last tier may be partly filled)
    DevBuf hot_ids;     // u32[Htot] x index of each hot source, by rank
    DevBuf hot_ent;     // u32[Mh]  hot edges, (bin, tier)-major: row_in_bin << 16 | index inside the tier; 0xFFFFFFFF = padding
    DevBuf hbin_v;      // u32[(B + G) T + 1] hot-edge range of each (bin, tier) (multiples of 4; of 512 with 2-byte records)
    // 2-byte hot records (hot16): hot_ent is u16[Mh], a record = row slot << 2 | how far its table index lies beyond the
    // record before it (0..3; slot Racc = a filler that only moves the index on); every 512 records — one wavefront's share
    // of a batch — have the index they start from in hot_base
    DevBuf hot_base;    // u16[Mh / 8 + 1]: the index the eight records of a lane start from
    uint32_t hot16 = 0;
    uint64_t Mh = 0;
    // host copies for launches over a range of source tiles / a group of bins (partitioned sweeps that
    // overlap the exchange of one part of x with the work on another)
    std::vector<uint32_t> wg_first_host; // u32[NT+1] first phase-1 workgroup of each tile
    std::vector<PbItem> items_host;      // the accumulate items in dispatch order
    int xcd_aware = 1;
};

namespace {

// ---- plan construction ---------------------------------------------------------------------------
// key = slot << (sb + bb + 1) | hot << (sb + bb) | bin << sb | src      (sb = bits of a source id, bb = of a bin id)
// Only the low sb + bb + 1 bits are sorted (LSD radix, stable): bin-major, then source, then CSR order — the
// accumulator slot rides in the unsorted high bits (its order inside a (bin, source) is irrelevant), which saves
// two of the seven 8-bit passes over the 8 GB of keys at scale 26.  A hot edge carries the hot index of its
// source in the source field and the flag bit above the bin, so hot keys sort behind every cold key, bin-major.
// A hub row (header) belongs to no ordinary bin: its bin field holds the VIRTUAL bin B + (its hub group) and its
// slot is its position inside the group.  Everything below simply sees B + G bins.
// `filter` (LDS; null for hub rows: every term of theirs must pass the value stream in source order): one bit per
// 2^fshift consecutive sources, set when one of them is hot.  It answers "not hot" for most edges without leaving the
// CU; only the rest look their source up in the rank table (134 MB at scale 26 — a random 2-byte gather per edge
// from that table was most of this kernel's time).
// hh_bit != 0 (a hub row walked by pb_hubseq_kernel): an edge from a hot source becomes a HOT key too, but keeps (virtual bin,
// SOURCE) as its sorted fields.  Ordinary hot keys carry ordinary bins (< B: no hub row uses the hot tables), so these sort
// behind all of them, in (group, source) order: no longer in the value stream, and where pb_hubseq_layout_kernel merges them
// with the group's cold entries into its blocks.
__device__ __forceinline__ uint64_t pb_make_key(uint64_t hi_cold, uint32_t src, int sb, int bb,
                                                const uint32_t *filter, int fshift,
                                                const uint32_t *__restrict__ hot_rank, uint64_t hh_bit = 0)
{
    if (filter) {
        const uint32_t blk = src >> fshift;
        if ((filter[blk >> 5] >> (blk & 31u)) & 1u) {
            const uint32_t h = hot_rank[src];
            if (h != 0xFFFFFFFFu)
                return hh_bit ? (hi_cold | (1ull << (sb + bb)) | src) : (hi_cold | (1ull << (sb + bb)) | h);
        }
    }
    return hi_cold | src;
}

// is any hub row's list out of ascending order?  one wavefront per row
__global__ void pb_hub_sorted_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt,
                                     const uint32_t *__restrict__ hub_rows, uint32_t n_hub, uint32_t *__restrict__ unsorted)
{
    const uint32_t lane = threadIdx.x & (kWave - 1), wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave,
                   nwaves = gridDim.x * blockDim.x / kWave;
    bool bad = false;
    for (uint32_t h = wave; h < n_hub; h += nwaves) {
        const uint32_t r = hub_rows[h], s = off[r], e = off[r + 1];
        for (uint32_t i = s + lane; i + 1u < e; i += kWave)
            bad |= tgt[i] > tgt[i + 1u];
    }
    if (__ballot(bad) && lane == 0)
        atomicOr(unsorted, 1u);
}

// gidx of row h's k-th in-neighbour (CSR order): the first entry of the row's group with that source.  `hubsrc` holds the source
// of every entry of the hub groups' part of the stream, non-decreasing inside a group, padding entries as the source before them
// — so the first entry with a given source is a real one, and it holds that source's out_score after the bin kernel.
__global__ __launch_bounds__(256) void pb_hubcsr_index_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt,
                                                              const uint32_t *__restrict__ hub_rows,
                                                              const uint32_t *__restrict__ g0, const uint32_t *__restrict__ rq0,
                                                              const uint32_t *__restrict__ rq1,
                                                              const uint32_t *__restrict__ hubsrc, uint32_t hub_q0, uint32_t n_hub,
                                                              uint32_t *__restrict__ gidx, uint32_t *__restrict__ missing)
{
    for (uint32_t h = blockIdx.x; h < n_hub; h += gridDim.x) {
        const uint32_t r = hub_rows[h], s = off[r], deg = off[r + 1] - s, q0 = rq0[h], q1 = rq1[h], base = g0[h];
        const uint32_t *cs = hubsrc + (q0 - hub_q0);
        const uint32_t padded = (deg + 3u) & ~3u;
        for (uint32_t k = threadIdx.x; k < padded; k += blockDim.x) {
            uint32_t at = q0;
            if (k < deg) {
                const uint32_t src = tgt[s + k];
                const uint32_t i = (uint32_t)lower_bound_fn(0, q1 - q0, src, [&](uint64_t j) { return (uint64_t)cs[j]; });
                if (i >= q1 - q0 || cs[i] != src)
                    atomicAdd(missing, 1u);
                else
                    at = q0 + i;
            }
            gidx[base + k] = at;
        }
    }
}

// hub rows: >= hub_deg in-edges
__global__ void pb_hubflag_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t hub_deg, uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride)
        flag[r] = (r < n && hub_deg && off[r + 1] - off[r] >= hub_deg) ? 1u : 0u;
}

// ... and rows BELOW it that sum many EQUAL terms (round 6; whole graphs: a row's sources are rows of the same CSR).  A source without
// in-edges carries (1 - d) / n in every sweep — the out_scores of such sources are equal within an out-degree class —, and sources with
// ONE in-edge from the same node (the pages of a site that only its front page links to) carry equal scores as well; the
// reference's left-to-right f32 sum of equal terms drifts SYSTEMATICALLY (every add rounds the same way while the sum stays in one
// binade): an exactly rounded sum misses the reference by that drift — by up to 6e-5 on a row of 4000 leaf followers, for most n that
// are not powers of two once a row has 2000 of them, never with 500 (tests/test_gpu_hub_order.py).  A row with at least
// `leaf_t` sources that have AT MOST ONE in-edge (GM_PB_HUB_LEAVES, default 512) is therefore a hub row whatever its length: summed
// the reference's way.  (RMAT: rows below 4096 in-edges have at most 92 of them at scale 22 / 24 / 26 — no BASELINE row is flagged;
// tools/leaf_sources_count.py; 7 ms of the plan build at scale 26.)  One wavefront looks at
// 64 consecutive rows and walks the lists of those whose in-degree lies in [leaf_t, hub_deg).
// (src_flags: a partition slice's list of which entries of its x vector are such sources, gm_csr_set_source_flags; null: whole graph)
__global__ __launch_bounds__(256) void pb_leafflag_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ src, uint32_t n,
                                                          uint32_t hub_deg, uint32_t leaf_t, const uint8_t *__restrict__ src_flags,
                                                          uint64_t src_flags_len, uint32_t *__restrict__ flag)
{
    const uint32_t lane = threadIdx.x & (kWave - 1u);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave, waves = (uint64_t)gridDim.x * blockDim.x / kWave;
    for (uint64_t r0 = wave * kWave; r0 < n; r0 += waves * kWave) {
        const uint32_t r = (uint32_t)r0 + lane;
        const uint32_t b = r < n ? off[r] : 0u, e = r < n ? off[r + 1] : 0u;
        uint64_t todo = __ballot(e - b >= leaf_t && e - b < hub_deg);
        while (todo) {
            const int l = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t rb_ = __shfl(b, l), re_ = __shfl(e, l);
            uint32_t c = 0;
            for (uint32_t k = rb_ + lane; k < re_; k += kWave) {
                const uint32_t s_ = src[k];
                if (src_flags)
                    c += (s_ < src_flags_len && src_flags[s_]) ? 1u : 0u;
                else
                    c += (s_ < n && off[s_ + 1] - off[s_] <= 1u) ? 1u : 0u;
            }
            const uint32_t total = (uint32_t)wave_sum((uint64_t)c);
            if (lane == 0 && total >= leaf_t)
                flag[(uint32_t)r0 + (uint32_t)l] = 1u;
        }
    }
}

// their ids and in-degrees in ascending row order (pos_h = exclusive scan of the hub flags)
__global__ void pb_hub_rows_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ pos_h, const uint32_t *__restrict__ flag,
                                   uint32_t n, uint32_t *__restrict__ hub_rows, uint32_t *__restrict__ hub_degs)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        if (flag[r]) {
            hub_rows[pos_h[r]] = r;
            hub_degs[pos_h[r]] = off[r + 1] - off[r];
        }
    }
}

// accumulator slots of the ordinary bins: rows with in-edges that are not hub rows, numbered consecutively
// pos_h[hub_rows[k]] = k: the hub rows' index space after pb_build has regrouped it
__global__ void pb_hub_repos_kernel(const uint32_t *__restrict__ hub_rows, uint32_t n_hub, uint32_t *__restrict__ pos_h)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_hub; k += stride)
        pos_h[hub_rows[k]] = k;
}

// (in place: flag holds the hub flags and receives "a row with in-edges that is no hub row")
__global__ void pb_rowflag_kernel(const uint32_t *__restrict__ off, uint32_t n, uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride) {
        const uint32_t deg = r < n ? off[r + 1] - off[r] : 0u;
        flag[r] = (deg && !(r < n && flag[r])) ? 1u : 0u;
    }
}

// (ordinary: pb_rowflag_kernel's flags — a row with in-edges that is not ordinary is a hub row)
__global__ void pb_cidx_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ ordinary,
                               uint32_t n, int rb, uint16_t *__restrict__ cidx, uint32_t *__restrict__ bin_rows)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        const uint32_t base = pos[(r >> rb) << rb];
        const uint32_t deg = off[r + 1] - off[r];
        cidx[r] = !deg ? PB_NULL : !ordinary[r] ? PB_HUBROW : (uint16_t)(pos[r] - base);
        if ((r & ((1u << rb) - 1u)) == 0) {
            const uint64_t end = ((uint64_t)((r >> rb) + 1) << rb);
            bin_rows[r >> rb] = pos[end < n ? end : n] - base;
        }
    }
}

// occurrences of every source among (a sample of) the edges: every `step`-th edge is counted — the H most
// frequent sources of a 1/8 sample are the same hubs, and any choice of hot set is correct
__global__ void pb_count_sources_kernel(const uint32_t *__restrict__ tgt, uint32_t m, uint32_t step,
                                        uint32_t *__restrict__ cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * step;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * step; i < m; i += stride)
        atomicAdd(&cnt[tgt[i]], 1u);
}

// candidates for the hot set: sources counted at least twice, as keys that sort ASCENDING into (count descending, id
// ascending) — ~count << 32 | id — appended in any order (the keys are distinct, the sorted order does not depend on it)
__global__ __launch_bounds__(256) void pb_count_keys_kernel(const uint32_t *__restrict__ cnt, uint32_t x_len,
                                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ n_keys)
{
    // every wavefront owns one contiguous stretch of the sources: count its candidates, reserve their places with
    // ONE atomic (a million wavefront-level adds on the same counter serialise at ~12 ns each), then write them
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per = ((x_len + nwaves - 1) / nwaves + kWave - 1) / kWave * kWave;
    const uint64_t lo = (uint64_t)wave * per;
    const uint64_t hi = lo + per < x_len ? lo + per : x_len;
    uint32_t mine = 0;
    for (uint64_t i = lo + lane; i < hi; i += kWave)
        mine += cnt[i] >= 2u ? 1u : 0u;
    const uint32_t total = (uint32_t)wave_sum((uint64_t)mine);
    if (total == 0)
        return;
    uint32_t at = 0;
    if (lane == 0)
        at = atomicAdd(n_keys, total);
    at = __shfl(at, 0, kWave);
    for (uint64_t base = lo; base < hi; base += kWave) {
        const uint64_t i = base + lane;
        const uint32_t c = i < hi ? cnt[i] : 0u;
        const uint64_t take = __ballot(c >= 2u);
        if (c >= 2u)
            keys[at + (uint32_t)__popcll(take & ((1ull << lane) - 1ull))] = ((uint64_t)(~c) << 32) | (uint32_t)i;
        at += (uint32_t)__popcll(take);
    }
}

// the first H candidates become hot: rank table + the block filter pb_keys_kernel stages in LDS
__global__ void pb_hot_select_kernel(const uint64_t *__restrict__ sorted, uint32_t H, uint32_t *__restrict__ hot_ids,
                                     uint32_t *__restrict__ hot_rank, uint32_t *__restrict__ hot_blk, int fshift)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H)
        return;
    const uint32_t id = (uint32_t)sorted[k];
    hot_ids[k] = id;
    hot_rank[id] = k;
    atomicOr(&hot_blk[(id >> fshift) >> 5], 1u << ((id >> fshift) & 31u));
}

// (bin, tier) of a hot key as one index: bin * T + rank / H
__device__ __forceinline__ uint32_t pb_hot_cell(uint64_t k, int sb, int bb, uint32_t H, uint32_t T)
{
    const uint32_t bin = (uint32_t)(k >> sb) & (uint32_t)((1ull << bb) - 1ull);
    const uint32_t r = (uint32_t)(k & ((1ull << sb) - 1ull));
    return bin * T + r / H;
}

// hstart[c] = first hot key (they are sorted by (bin, rank)) whose cell is >= c, for c in [0, cells]
__global__ void pb_hot_bounds_kernel(const uint64_t *__restrict__ hkeys, uint32_t mh, int sb, int bb, uint32_t H, uint32_t T,
                                     uint32_t cells, uint32_t *__restrict__ hstart)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= mh; i += stride) {
        const uint32_t lo = i == 0 ? 0u : pb_hot_cell(hkeys[i - 1], sb, bb, H, T) + 1u;
        const uint32_t hi = i == mh ? cells : pb_hot_cell(hkeys[i], sb, bb, H, T);
        for (uint32_t c = lo; c <= hi; ++c)
            hstart[c] = i;
    }
}

__global__ void pb_hot_fill_kernel(const uint64_t *__restrict__ hkeys, uint32_t mh, const uint32_t *__restrict__ hstart,
                                   const uint32_t *__restrict__ hbin_v, int sb, int bb, uint32_t H, uint32_t T,
                                   uint32_t *__restrict__ hot_ent)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < mh; i += stride) {
        const uint64_t k = hkeys[i];
        const uint32_t cell = pb_hot_cell(k, sb, bb, H, T);
        const uint32_t slot = (uint32_t)(k >> (sb + bb + 1));
        const uint32_t r = (uint32_t)(k & ((1ull << sb) - 1ull));
        hot_ent[hbin_v[cell] + (i - hstart[cell])] = (slot << 16) | (r % H);
    }
}

__global__ void pb_pad4_sizes_kernel(const uint32_t *__restrict__ start, uint32_t count, uint32_t *__restrict__ padded)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= count; b += stride)
        padded[b] = b == count ? 0u : ((start[b + 1] - start[b] + 3u) & ~3u);
}

// ---- 2-byte hot records -------------------------------------------------------------------------------------------------
// Inside a (bin, tier) cell the hot keys are sorted by table index, and a cell holds about as many records as the table has
// entries: the index of a record is that of the record before it plus 0..3 nearly always.  entries[i] = 1 + the fillers
// record i needs in front of it (each moves the index on by 3); the first record of a cell starts from its own index.
constexpr uint32_t PB_H16_CHUNK = 512; // a cell's records are padded to whole wavefront shares (eight per lane)
constexpr uint32_t PB_H16_LANE = 8;    // records with one base index: a lane's 16-byte load (a scan over the wavefront
                                       // instead — one base per 512 records — cost more than these 0.25 B per record save:
                                       // 2.93 against 2.81 ms per sweep at scale 26, tools/runs/r04_call56.sh)
__device__ __forceinline__ uint32_t pb_h16_gap(const uint64_t *__restrict__ hkeys, uint32_t i, int sb, int bb, uint32_t H, uint32_t T)
{
    const uint64_t k = hkeys[i];
    if (i == 0 || pb_hot_cell(hkeys[i - 1], sb, bb, H, T) != pb_hot_cell(k, sb, bb, H, T))
        return 0u;
    const uint32_t r = (uint32_t)(k & ((1ull << sb) - 1ull)), rp = (uint32_t)(hkeys[i - 1] & ((1ull << sb) - 1ull));
    return r % H - rp % H; // same cell: same tier, sorted by rank
}

__global__ void pb_h16_count_kernel(const uint64_t *__restrict__ hkeys, uint32_t mh, int sb, int bb, uint32_t H, uint32_t T,
                                    uint32_t *__restrict__ entries)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= mh; i += stride) {
        if (i == mh) {
            entries[i] = 0u;
            continue;
        }
        const uint32_t gap = pb_h16_gap(hkeys, i, sb, bb, H, T);
        entries[i] = 1u + (gap > 3u ? (gap - 1u) / 3u : 0u);
    }
}

// padded[c] = the cell's entries (records + fillers) rounded up to a whole chunk
__global__ void pb_h16_sizes_kernel(const uint32_t *__restrict__ hstart, const uint32_t *__restrict__ epos, uint32_t cells,
                                    uint32_t *__restrict__ padded)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= cells; c += stride)
        padded[c] = c == cells ? 0u : ((epos[hstart[c + 1]] - epos[hstart[c]] + PB_H16_CHUNK - 1u) & ~(PB_H16_CHUNK - 1u));
}

__global__ void pb_h16_pattern_kernel(uint16_t *__restrict__ out, uint64_t count, uint16_t value)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        out[i] = value;
}

__global__ void pb_h16_fill_kernel(const uint64_t *__restrict__ hkeys, uint32_t mh, const uint32_t *__restrict__ hstart,
                                   const uint32_t *__restrict__ epos, const uint32_t *__restrict__ hbin_v, int sb, int bb,
                                   uint32_t H, uint32_t T, uint32_t filler_slot, uint16_t *__restrict__ hot16,
                                   uint16_t *__restrict__ hot_base)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < mh; i += stride) {
        const uint64_t k = hkeys[i];
        const uint32_t cell = pb_hot_cell(k, sb, bb, H, T);
        const uint32_t slot = (uint32_t)(k >> (sb + bb + 1));
        const uint32_t idx = (uint32_t)(k & ((1ull << sb) - 1ull)) % H;
        const uint32_t gap = pb_h16_gap(hkeys, i, sb, bb, H, T);
        const uint32_t nf = gap > 3u ? (gap - 1u) / 3u : 0u;
        const uint32_t cell_at = hbin_v[cell], cell_end = hbin_v[cell + 1];
        uint32_t pos = cell_at + (epos[i] - epos[hstart[cell]]);
        uint32_t running = idx - gap; // the index the entries in front of this record start from
        if (i == hstart[cell])
            hot_base[cell_at / PB_H16_LANE] = (uint16_t)idx; // a cell's first record: gap 0 from its own index
        for (uint32_t f = 0; f < nf; ++f, ++pos) {
            running += 3u;
            hot16[pos] = (uint16_t)((filler_slot << 2) | 3u);
            if ((pos & (PB_H16_LANE - 1u)) == PB_H16_LANE - 1u && pos + 1u < cell_end)
                hot_base[(pos + 1u) / PB_H16_LANE] = (uint16_t)running;
        }
        hot16[pos] = (uint16_t)((slot << 2) | (idx - running));
        if ((pos & (PB_H16_LANE - 1u)) == PB_H16_LANE - 1u && pos + 1u < cell_end)
            hot_base[(pos + 1u) / PB_H16_LANE] = (uint16_t)idx;
    }
}

__global__ void pb_hot_gather_kernel(const float *__restrict__ x_in, const uint32_t *__restrict__ hot_ids, uint32_t H,
                                     float *__restrict__ hot_x)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < H)
        hot_x[k] = x_in[hot_ids[k]];
}

constexpr int PB_KEYS_BLOCK = 1024;
constexpr int PB_FILTER_BITS = 20; // the hot-source block filter has at most 2^20 bits (128 KiB of LDS)
constexpr int PB_FILTER_DEFAULT = 18; // measured at scale 26: 2^20 bits 14.2 ms, 2^19 12.1, 2^18 11.8, 2^17 12.4, 2^16 13.1

__global__ __launch_bounds__(PB_KEYS_BLOCK) void pb_keys_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tgt,
                                                               uint32_t n, int rb, int sb, int bb,
                                                               const uint32_t *__restrict__ hot_blk, uint32_t filter_words,
                                                               int fshift, const uint32_t *__restrict__ hot_rank,
                                                               const uint16_t *__restrict__ cidx,
                                                               const uint32_t *__restrict__ pos_h,
                                                               const uint32_t *__restrict__ hub_first, uint32_t B, uint32_t G,
                                                               const uint32_t *__restrict__ group_long, uint64_t hh_bit,
                                                               uint64_t *__restrict__ keys)
{
    extern __shared__ uint32_t pb_filter[];
    for (uint32_t i = threadIdx.x; i < filter_words; i += PB_KEYS_BLOCK)
        pb_filter[i] = hot_blk[i];
    __syncthreads();
    const uint32_t *filter = hot_blk ? pb_filter : nullptr;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_pad = (n + kWave - 1) / kWave * kWave;
    const uint32_t rmask = (1u << rb) - 1u;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_pad; r += stride) {
        uint32_t s = 0, e = 0;
        if (r < n) {
            s = off[r];
            e = off[r + 1];
        }
        uint32_t slot = r < n ? cidx[r] : 0u; // rows with edges always own a slot
        uint32_t vbin = r >> rb;
        const uint32_t len = e - s;
        const bool hub = r < n && len && slot == PB_HUBROW;
        int hot_hub = 0; // a hub row whose hot terms leave the value stream (not a long row, and the plan wants it: hh_bit)
        if (hub) { // virtual bin B + group, slot = position inside the group
            const uint32_t i = pos_h[r];
            const uint32_t g = (uint32_t)lower_bound_fn(0, G + 1, (uint64_t)i + 1, [&](uint64_t k) { return (uint64_t)hub_first[k]; }) - 1u;
            vbin = B + g;
            slot = i - hub_first[g];
            hot_hub = hh_bit && !group_long[g] ? 1 : 0;
        }
        const uint64_t hi = ((uint64_t)(slot & rmask) << (sb + bb + 1)) | ((uint64_t)vbin << sb);
        if (len <= 32)
            for (uint32_t i = s; i < e; ++i)
                keys[i] = pb_make_key(hi, tgt[i], sb, bb, hub ? nullptr : filter, fshift, hot_rank);
        uint64_t big = __ballot(len > 32);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t bs = __shfl(s, src, kWave), be = __shfl(e, src, kWave);
            const uint64_t bhi = __shfl(hi, src, kWave);
            const bool bhub = __shfl((int)hub, src, kWave) != 0;
            const uint64_t bhh = __shfl(hot_hub, src, kWave) ? hh_bit : 0ull;
            // four loads in flight per lane: one at a time left the long rows (most of the edges) waiting on each
            for (uint32_t i = bs + lane; i < be; i += kWave * 4u) {
                uint32_t t4[4];
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q)
                    t4[q] = i + q * kWave < be ? tgt[i + q * kWave] : 0u;
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q)
                    if (i + q * kWave < be)
                        keys[i + q * kWave] = pb_make_key(bhi, t4[q], sb, bb, (bhub && !bhh) ? nullptr : filter, fshift, hot_rank, bhh);
            }
        }
    }
}

__device__ __forceinline__ uint64_t pb_seg_of_key(uint64_t k, int bb, int sb, int s_log)
{
    // (bin, tile) as one comparable integer: bin << 32 | tile
    const uint64_t bin = (k >> sb) & ((1ull << bb) - 1ull);
    const uint64_t src = k & ((1ull << sb) - 1ull);
    return (bin << 32) | (src >> s_log);
}

__global__ void pb_flags_kernel(const uint64_t *__restrict__ keys, uint32_t m, int bb, int sb, int s_log,
                                uint32_t *__restrict__ flag)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) {
        const uint64_t k = keys[q];
        flag[q] = (q == 0 || pb_seg_of_key(keys[q - 1], bb, sb, s_log) != pb_seg_of_key(k, bb, sb, s_log)) ? 1u : 0u;
    }
}

// segid = inclusive_scan(flag); for every segment start: vstart[j] = q, segbin[j] = bin, and the key that sorts the
// segments into phase-1 order with their bin-major index as payload: tile << (bb + jb) | bin << jb | j
__global__ void pb_segments_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ flag,
                                   const uint32_t *__restrict__ segid_incl, uint32_t m, int bb, int sb, int s_log, int jb,
                                   uint32_t *__restrict__ vstart, uint64_t *__restrict__ segkey,
                                   uint64_t *__restrict__ segbin)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride)
        if (flag[q]) {
            const uint32_t j = segid_incl[q] - 1;
            const uint64_t bt = pb_seg_of_key(keys[q], bb, sb, s_log);
            vstart[j] = q;
            segkey[j] = ((bt & 0xFFFFFFFFull) << (bb + jb)) | ((bt >> 32) << jb) | j;
            segbin[j] = bt >> 32;
        }
}

// the payload of the sorted segment keys: segval[r] = bin-major index of the r-th segment in phase-1 order
__global__ void pb_segval_kernel(const uint64_t *__restrict__ segkey_sorted, uint32_t NS, int jb, uint32_t *__restrict__ segval)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < NS; r += stride)
        segval[r] = (uint32_t)(segkey_sorted[r] & ((1ull << jb) - 1ull));
}

// segments in phase-1 order (rank r): cnt[r]; tile_seg[t] = first rank of tile t
// padded (multiple of 4) segment sizes: cnt_p1[r] in phase-1 order, cnt_v[j] in bin-major order
__global__ void pb_seg_counts_kernel(const uint32_t *__restrict__ segval_sorted, const uint32_t *__restrict__ vstart,
                                     uint32_t NS, uint32_t m, uint32_t *__restrict__ cnt_p1, uint32_t *__restrict__ cnt_v,
                                     uint32_t pad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r <= NS; r += stride) {
        if (r == NS) {
            cnt_p1[r] = 0;
            cnt_v[r] = 0;
            continue;
        }
        const uint32_t j = segval_sorted[r];
        const uint32_t end = j + 1 < NS ? vstart[j + 1] : m;
        cnt_p1[r] = (end - vstart[j] + pad - 1u) & ~(pad - 1u);
        const uint32_t end2 = r + 1 < NS ? vstart[r + 1] : m; // r doubles as a bin-major index here
        cnt_v[r] = (end2 - vstart[r] + pad - 1u) & ~(pad - 1u);
    }
}

// padded tile sizes from the exclusive scan `cs` of cnt and the tile boundaries
__global__ void pb_tile_sizes_kernel(const uint32_t *__restrict__ tile_seg, const uint32_t *__restrict__ cs, uint32_t NT,
                                     uint32_t *__restrict__ tile_pad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t <= NT; t += stride) {
        if (t == NT) {
            tile_pad[t] = 0;
            continue;
        }
        const uint32_t c = cs[tile_seg[t + 1]] - cs[tile_seg[t]];
        tile_pad[t] = (c + PB_WBLK - 1u) & ~(PB_WBLK - 1u);
    }
}

// A tile's phase-1 range is rounded up to 256 entries; the entries behind its last segment count as
// padding of that segment (the bin kernel has no "nothing here" marker), so the segment's slice of the
// value stream grows by the same amount.
__global__ void pb_tile_tail_kernel(const uint32_t *__restrict__ tile_seg, const uint32_t *__restrict__ cs,
                                    const uint32_t *__restrict__ tile_pad, const uint32_t *__restrict__ segval_sorted,
                                    uint32_t NT, uint32_t *__restrict__ cnt_v)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < NT; t += stride) {
        const uint32_t r0 = tile_seg[t], r1 = tile_seg[t + 1];
        if (r1 > r0)
            cnt_v[segval_sorted[r1 - 1]] += tile_pad[t] - (cs[r1] - cs[r0]);
    }
}

// per phase-1 segment r: pstart, delta; and the inverse permutation rank_of[j] = r
__global__ void pb_seg_layout_kernel(const uint64_t *__restrict__ segkey_sorted, const uint32_t *__restrict__ segval_sorted,
                                     const uint32_t *__restrict__ vstart4, const uint32_t *__restrict__ cs,
                                     const uint32_t *__restrict__ tile_seg, const uint32_t *__restrict__ tile_p,
                                     uint32_t NS, int tile_shift, uint32_t *__restrict__ pstart,
                                     uint32_t *__restrict__ delta, uint32_t *__restrict__ rank_of)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < NS; r += stride) {
        const uint32_t t = (uint32_t)(segkey_sorted[r] >> tile_shift);
        const uint32_t j = segval_sorted[r];
        const uint32_t ps = tile_p[t] + (cs[r] - cs[tile_seg[t]]);
        pstart[r] = ps;
        delta[r] = vstart4[j] - ps;
        rank_of[j] = r;
    }
}

__global__ void pb_fill_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ segid_incl,
                               const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ vstart4,
                               const uint32_t *__restrict__ rank_of, const uint32_t *__restrict__ pstart, uint32_t m,
                               int bb, int sb, int s_log, uint16_t *__restrict__ p1_src, uint16_t *__restrict__ p2_dst,
                               uint32_t *__restrict__ hubsrc, uint32_t hub_q0)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) {
        const uint32_t j = segid_incl[q] - 1;
        const uint32_t vs = vstart[j];
        const uint64_t k = keys[q];
        const uint32_t p = pstart[rank_of[j]] + (q - vs);
        const uint32_t src = (uint32_t)(k & ((1ull << sb) - 1ull));
        p1_src[p] = (uint16_t)((src & ((1u << s_log) - 1u)) | (q == vs ? PB_FLAG : 0));
        p2_dst[vstart4[j] + (q - vs)] = (uint16_t)(k >> (sb + bb + 1));
        if (hubsrc && vstart4[j] >= hub_q0) // a hub group's entry (the hub groups are the stream's last bins): its source
            hubsrc[vstart4[j] + (q - vs) - hub_q0] = src;
    }
}

// padding entries of the hub groups' part of the stream (0xFFFFFFFF after the fill) take the source of the entry before them:
// the part reads as a non-decreasing sequence per group, which is what the merge with the hot records goes by
__global__ void pb_hubsrc_pad_kernel(uint32_t *__restrict__ hubsrc, uint32_t count)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        if (hubsrc[i] == 0xFFFFFFFFu) {
            uint32_t j = i;
            while (j > 0 && hubsrc[j - 1] == 0xFFFFFFFFu)
                --j;
            // (a padding run is at most 3 + 255 entries; other threads may be filling it from the same source meanwhile: same value)
            const uint32_t v = j > 0 ? hubsrc[j - 1] : 0u;
            hubsrc[i] = v == 0xFFFFFFFFu ? 0u : v;
        }
}

// GM_PB_BIN_GAP (measurement): a pseudo-random run of unused entries (multiple of 4, below `max_gap`) behind every bin's
// last segment, so that the bins' areas of the value stream do not start at regular address intervals
__global__ void pb_bin_gap_kernel(const uint32_t *__restrict__ bin_seg, uint32_t B, uint32_t max_gap,
                                  uint32_t *__restrict__ cnt_v)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += stride)
        if (bin_seg[b + 1] > bin_seg[b])
            cnt_v[bin_seg[b + 1] - 1u] += ((b * 2654435761u) >> 12) % max_gap & ~3u;
}

// bin_v[b] = padded value-stream position of the first segment of a bin >= b
__global__ void pb_bin_ranges_kernel(const uint32_t *__restrict__ bin_seg, const uint32_t *__restrict__ vstart4, uint32_t B,
                                     uint32_t *__restrict__ bin_v)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= B; b += stride)
        bin_v[b] = vstart4[bin_seg[b]];
}

// phase-1 workgroups: tile t is split into ceil(len / PB_BIN_CHUNK) chunks
__global__ void pb_wg_count_kernel(const uint32_t *__restrict__ tile_p, uint32_t NT, uint32_t chunk,
                                   uint32_t *__restrict__ wg_cnt)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t <= NT; t += stride)
        wg_cnt[t] = t == NT ? 0u : (tile_p[t + 1] - tile_p[t] + chunk - 1u) / chunk;
}

// work items in tile-major order: wg_tile[w] / wg_p0[w] of item w
__global__ void pb_wg_fill_kernel(const uint32_t *__restrict__ tile_p, const uint32_t *__restrict__ wg_first, uint32_t NT,
                                  uint32_t PB_BIN_CHUNK, uint32_t NW, uint32_t *__restrict__ wg_tile,
                                  uint32_t *__restrict__ wg_p0)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < NW; w += stride) {
        // tile of item w: last t with wg_first[t] <= w
        uint32_t lo = 0, hi = NT;
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (wg_first[mid] <= w)
                lo = mid;
            else
                hi = mid;
        }
        wg_tile[w] = lo;
        wg_p0[w] = tile_p[lo] + (w - wg_first[lo]) * PB_BIN_CHUNK;
    }
}

// Launch slot s runs on XCD s % 8 (observed dispatch rule; used for speed only).  The slots of one XCD
// take consecutive work items (tile-major order), so the cache lines shared by the adjacent segments
// of consecutive tiles in the value stream are completed inside one L2 instead of leaving two partial
// write-backs.  slot -> item is a bijection on [0, count): item = base[s % 8] + s / 8.
__device__ __forceinline__ uint32_t pb_xcd_item(uint32_t s, uint32_t count)
{
    const uint32_t x = s & 7u;
    uint32_t base = 0;
    for (uint32_t y = 0; y < x; ++y)
        base += (count - y + 7u) / 8u;
    return base + (s >> 3);
}

// chunk_seg[c] = number of segments with pstart < 256 c
__global__ void pb_chunk_seg_kernel(const uint32_t *__restrict__ pstart, uint32_t NS, uint32_t nchunks,
                                    uint32_t *__restrict__ chunk_seg)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += stride) {
        const uint64_t target = (uint64_t)c * PB_WBLK;
        chunk_seg[c] = (uint32_t)lower_bound_fn(0, NS, target, [&](uint64_t r) { return (uint64_t)pstart[r]; });
    }
}

// boundaries of a sorted u64 key array after a shift: out[v] = first index whose key >> shift is >= v
__global__ void pb_bounds_kernel(const uint64_t *__restrict__ keys, uint32_t count, int shift, uint32_t nvals,
                                 uint32_t *__restrict__ out, uint32_t mask = 0xFFFFFFFFu)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= count; i += stride) {
        const uint32_t lo = i == 0 ? 0u : ((uint32_t)(keys[i - 1] >> shift) & mask) + 1u;
        const uint32_t hi = i == count ? nvals : ((uint32_t)(keys[i] >> shift) & mask);
        for (uint32_t v = lo; v <= hi; ++v)
            out[v] = i;
    }
}

// ---- the sweep -----------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct alignas(8) U16x4 {
    uint16_t a, b, c, d;
};

// A workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for every outstanding global load
// (s_waitcnt vmcnt(0)): a kernel that keeps the next block's loads in flight across its barriers would wait for them at
// the first one — measured on pb_hubseq_kernel / pb_hublong_kernel: the whole memory latency exposed once per block.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int PB_BIN_U = 4;         // 256-entry blocks a wavefront of pb_bin_kernel handles per pipeline stage
constexpr int PB_ACC_DEPTH = 4;     // register groups (float4 + 4 slots per lane) of pb_accum_kernel's value stream in flight
constexpr int PB_EPI = 2;           // groups of 4 rows a lane of the accumulate epilogue keeps in flight
constexpr uint32_t PB_DCACHE = 4096; // segment deltas cached in LDS per workgroup (16 KiB)

// ABL != 0 are measurement-only variants (GM_PB_ABLATE, wrong results by design): they remove one
// ingredient at a time to show what bounds the kernel.  bin: 1 = no LDS gather, 3 = no stores.  accumulate: 3 = no epilogue, 4 = no streaming loops
// (removing the LDS atomics or making them conflict-free changed nothing: measured, then deleted).  bin: 4 = no x tile load,
// 5 = x tile load only.
template <int ABL, int S_LOG>
__global__ __launch_bounds__(PB_BIN_BLOCK) void pb_bin_kernel(const float *__restrict__ x_in, uint64_t x_len,
                                                              const uint32_t *__restrict__ tile_p,
                                                              const uint32_t *__restrict__ wg_tile,
                                                              const uint32_t *__restrict__ wg_p0,
                                                              const uint16_t *__restrict__ p1_src,
                                                              const uint32_t *__restrict__ chunk_seg,
                                                              const uint32_t *__restrict__ delta, float *__restrict__ vals,
                                                              uint32_t PB_BIN_CHUNK, uint32_t w_first, int xcd_aware,
                                                              const uint32_t *__restrict__ item_list,
                                                              const uint32_t *__restrict__ hot_ids, uint32_t Htot,
                                                              float *__restrict__ hot_x)
{
    // whole sweeps: the out_scores of the hot sources (read by the accumulate kernel from hot_x) are gathered here, a share
    // per workgroup, instead of by a launch of their own in front of this one (pb_hot_gather_kernel: 4 us + a kernel boundary)
    if (hot_ids) {
        const uint32_t per = (Htot + gridDim.x - 1u) / gridDim.x;
        const uint32_t k = blockIdx.x * per + threadIdx.x;
        if (threadIdx.x < per && k < Htot)
            hot_x[k] = x_in[hot_ids[k]];
    }
    constexpr uint32_t PB_S = 1u << S_LOG;                          // sources per tile
    extern __shared__ float xs[];                                   // PB_S floats ...
    uint32_t *dl = reinterpret_cast<uint32_t *>(xs + PB_S);         // ... + PB_DCACHE segment deltas
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = xcd_aware ? pb_xcd_item(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint32_t item = item_list ? item_list[slot] : w_first + slot; // a list: the items of a region of several tile ranges
    const uint32_t t = wg_tile[item];
    const uint32_t p_begin = wg_p0[item];
    const uint32_t tile_end = tile_p[t + 1];
    const uint32_t p_end = (tile_end - p_begin) < PB_BIN_CHUNK ? tile_end : p_begin + PB_BIN_CHUNK;
    const uint64_t x0 = (uint64_t)t * PB_S;
    const uint32_t xn = (uint32_t)((x_len - x0) < PB_S ? (x_len - x0) : PB_S);
    constexpr int U = PB_BIN_U;
    constexpr uint32_t STEP = PB_BIN_BLOCK * PB_VEC; // entries per workgroup step (4096)
    float sink = 0.f;
    // every wavefront covers one aligned 256-entry block per step: lane l owns entries 4l..4l+3.
    // Software-pipelined: the ids of step group i+1 are requested BEFORE the stores of group i are issued.
    // Memory operations of a wavefront retire in order (one vmcnt), so loads issued after the stores
    // could only be consumed once those stores have drained — measured as load and store time adding
    // up (0.57 + 0.94 ms at scale 26) instead of overlapping.
    U16x4 v[U];
    uint32_t cs[U];
    auto fetch = [&](uint32_t p0, U16x4(&vv)[U], uint32_t(&cc)[U]) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t p = p0 + k * STEP;
            const bool in = p < p_end; // uniform per wavefront: ranges are multiples of 256
            if (in) {
                const u32x2 raw = *reinterpret_cast<const u32x2 *>(p1_src + p);
                vv[k] = U16x4{(uint16_t)raw.x, (uint16_t)(raw.x >> 16), (uint16_t)raw.y, (uint16_t)(raw.y >> 16)};
            } else {
                vv[k] = U16x4{0, 0, 0, 0};
            }
            cc[k] = in ? chunk_seg[p / PB_WBLK] : 0u;
        }
    };
    const uint32_t p_first = p_begin + tid * PB_VEC;
    if (ABL != 5 && p_first < p_end)
        fetch(p_first, v, cs);
    if (ABL == 4) {
    } else if ((xn & 3u) == 0 && ((x0 & 3u) == 0)) {
        const float4 *src4 = reinterpret_cast<const float4 *>(x_in + x0);
        float4 *dst4 = reinterpret_cast<float4 *>(xs);
        for (uint32_t i = tid; i < xn / 4; i += PB_BIN_BLOCK)
            dst4[i] = src4[i];
    } else {
        for (uint32_t i = tid; i < xn; i += PB_BIN_BLOCK)
            xs[i] = x_in[x0 + i];
    }
    // segments this workgroup can touch: ranks [r_lo, r_hi)
    const uint32_t cs_first = chunk_seg[p_begin / PB_WBLK];
    const uint32_t r_lo = cs_first ? cs_first - 1u : 0u;
    const uint32_t r_hi = chunk_seg[p_end / PB_WBLK];
    const uint32_t r_cached = (r_hi - r_lo) < PB_DCACHE ? (r_hi - r_lo) : PB_DCACHE;
    for (uint32_t i = tid; i < r_cached; i += PB_BIN_BLOCK)
        dl[i] = delta[r_lo + i];
    __syncthreads();
    const uint32_t lane = tid & (kWave - 1);
    const uint64_t le_mask = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    for (uint32_t p0 = p_first; p0 < (ABL == 5 ? p_begin : p_end); p0 += STEP * U) {
        U16x4 vn[U];
        uint32_t cn[U];
        const uint32_t pn = p0 + STEP * U;
        if (pn < p_end)
            fetch(pn, vn, cn);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const uint32_t p = p0 + k * STEP;
            const bool valid = p < p_end; // every entry of the range belongs to a segment (padding included)
            const uint64_t starts = __ballot(valid && (v[k].a & PB_FLAG));
            if (valid) {
                const uint32_t rank = cs[k] + (uint32_t)__popcll(starts & le_mask) - 1u;
                const uint32_t ri = rank - r_lo;
                const uint32_t dlt = ri < PB_DCACHE ? dl[ri] : delta[rank];
                f32x4 o;
                if (ABL == 1) {
                    o.x = (float)v[k].a, o.y = (float)v[k].b, o.z = (float)v[k].c, o.w = (float)v[k].d;
                } else {
                    o.x = xs[v[k].a & (PB_S - 1u)];
                    o.y = xs[v[k].b & (PB_S - 1u)];
                    o.z = xs[v[k].c & (PB_S - 1u)];
                    o.w = xs[v[k].d & (PB_S - 1u)];
                }
                // padding lanes write padding slots
                if (ABL == 3)
                    sink += o.x + o.y + o.z + o.w + (float)dlt;
                else
                    *reinterpret_cast<f32x4 *>(vals + (p + dlt)) = o;
            }
        }
        if (pn < p_end) {
#pragma unroll
            for (int k = 0; k < U; ++k) {
                v[k] = vn[k];
                cs[k] = cn[k];
            }
        }
    }
    if (ABL == 5)
        sink = xs[(tid * 17u) & (PB_S - 1u)];
    if ((ABL == 3 || ABL == 5) && sink == 12345.678f)
        vals[tid] = sink;
}

// The sweep's error WITHOUT a kernel of its own (round 6; pb_err_kernel + the gap in front of it were 15-20 us of a 200 us sweep
// at scale 22): every workgroup of the accumulate launch and of the hub launches writes its error slot write-through, waits
// for that store, and draws a ticket; the workgroup that draws the last one sums the slots — by its first 256 threads, slot b
// in lane b mod 256, a wavefront butterfly, then the four wavefronts' sums left to right: the same order whichever kernel's
// workgroup happens to be last, so the value does not depend on the schedule.  (pb_sweep_accum_part's pieces keep the kernel:
// their launches are the caller's to count.)
__device__ __forceinline__ void pb_err_fold(const PbErrFold &f)
{
    if (!f.ctr) // (a kernel argument: uniform)
        return;
    __shared__ double fold_red[4];
    __shared__ uint32_t fold_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's error store, if it made one, has left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = atomicAdd(f.ctr, 1u);
        fold_last = prev + 1u == f.total ? 1u : 0u;
        if (fold_last)
            st_agent(f.ctr, 0u); // ready for the next sweep (which starts behind this launch and the hub launches' join)
    }
    __syncthreads();
    if (!fold_last)
        return;
    double a = 0.0;
    if (threadIdx.x < 256u) {
        for (uint32_t b = threadIdx.x; b < f.count; b += 256u)
            a += ld_agent(f.slots + b);
        a = wave_sum(a);
        if ((threadIdx.x & (kWave - 1)) == 0)
            fold_red[threadIdx.x / kWave] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        *f.out = ((fold_red[0] + fold_red[1]) + fold_red[2]) + fold_red[3];
}

__device__ __forceinline__ unsigned long long pb_to_fix(float x)
{
    return (unsigned long long)(x * PB_FIX_SCALE); // exact scaling by 2^62, truncation below 2^-62
}

// BF (measurement, GM_PB_ACC_BRANCHFREE=1): padding entries go to one more accumulator behind the rows' instead of being
// branched around (a compare, a scalar mask save / restore and a branch for each of the 1.1 G entries of a sweep at scale 26).
// Measured in alternating fresh processes on one box (tools/runs/r04_call25.sh): SLOWER, 1831 against 1620 us — the 2 % of
// padding entries then meet on ONE LDS address and same-address atomics serialise.
template <int ABL, int D = PB_ACC_DEPTH, bool BF = false, bool H16 = false>
__global__ __launch_bounds__(PB_ACC_BLOCK) void pb_accum_kernel(const float *__restrict__ vals,
                                                                const uint16_t *__restrict__ p2_dst,
                                                                const PbItem *__restrict__ items,
                                                                const uint32_t *__restrict__ hot_ent,
                                                                const uint32_t *__restrict__ hbin_v,
                                                                const float *__restrict__ hot_x, uint32_t H, uint32_t T,
                                                                uint32_t Htot, unsigned long long *partials, uint32_t *tickets,
                                                                const uint16_t *__restrict__ cidx,
                                                                const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                                                float *__restrict__ x_out, double *__restrict__ bin_err,
                                                                uint32_t n_local, uint32_t R, uint32_t Racc, float base,
                                                                float damping, const uint16_t *__restrict__ hot_base, PbErrFold fold)
{
    extern __shared__ unsigned long long acc[]; // Racc fixed-point sums (one per ordinary row WITH in-edges)
    __shared__ double red[PB_ACC_BLOCK / kWave];
    __shared__ bool is_last;
    const PbItem item = items[blockIdx.x]; // longest items are dispatched first
    const uint32_t b = item.bin, tid = threadIdx.x;
    float *hot = reinterpret_cast<float *>(acc + Racc + 2); // the out_scores of one tier of hot sources (H at most); acc[Racc]: padding
    const uint32_t qb = item.q0, qe = (ABL == 4 || ABL == 5 ? item.q0 : item.q1); // multiples of 4 (ABL 5: hot edges only, 6: stream only)
    constexpr uint32_t STEP = PB_ACC_BLOCK * PB_VEC;
    // Every phase keeps several independent loads per lane in flight and the first group of the value
    // stream is requested before the prologue: with one or two workgroups per CU nothing else hides a
    // phase that waits for one load at a time (measured at scale 26: hot table 27 + hot edges 18 +
    // epilogue 32 dependent round trips per workgroup were a third of the kernel).
    // The value stream runs through a RING of D register groups: a group is consumed and at once refilled with the entries
    // D steps ahead, so D groups (D x 24 bytes per lane, 96 KiB per workgroup) are in flight all the time — the earlier
    // "fetch the next pair, consume this pair, copy" kept two in flight with the same number of registers, and with one
    // workgroup per CU the bytes in flight are what the CU's share of the HBM bandwidth is made of.
    f32x4 v[D];
    U16x4 d[D];
    auto fetch1 = [&](uint32_t q, f32x4 &vv, U16x4 &dd) {
        if (q < qe) {
            vv = *reinterpret_cast<const f32x4 *>(vals + q);
            const u32x2 raw = *reinterpret_cast<const u32x2 *>(p2_dst + q);
            dd = U16x4{(uint16_t)raw.x, (uint16_t)(raw.x >> 16), (uint16_t)raw.y, (uint16_t)(raw.y >> 16)};
        } else {
            dd = U16x4{PB_NULL, PB_NULL, PB_NULL, PB_NULL};
        }
    };
    const uint32_t q_first = qb + tid * PB_VEC;
#pragma unroll
    for (int k = 0; k < D; ++k)
        fetch1(q_first + (uint32_t)k * STEP, v[k], d[k]);
    for (uint32_t i = tid; i < Racc; i += PB_ACC_BLOCK)
        acc[i] = 0ull;
    // hot edges of this item, tier by tier: [ha, hb) inside the (bin, tier) cell; a slice of an over-long bin takes its
    // share of every cell
    auto tier_range = [&](uint32_t t, uint32_t &ha, uint32_t &hb) {
        ha = hbin_v[b * T + t];
        hb = hbin_v[b * T + t + 1];
        if (item.nparts > 1) {
            constexpr uint32_t GRAN = H16 ? PB_H16_CHUNK : 4u; // (2-byte records: whole chunks, they share a base index)
            const uint32_t per = (((hb - ha) + item.nparts - 1u) / item.nparts + GRAN - 1u) & ~(GRAN - 1u);
            const uint32_t lo = ha + item.part * per;
            ha = lo < hb ? lo : hb;
            hb = (hb - ha) < per ? hb : ha + per;
        }
    };
    // Two tables in LDS (T > 1): while the edges of one tier gather from one of them, the next tier's table is on its
    // way through registers (tier_fetch, three float4 per lane) into the other (tier_store), and ONE barrier per tier
    // separates the two uses of a buffer.  A tier switch that waited for its table cost every workgroup ~5 us
    // (measured: +0.04 ms per tier and sweep at scale 26).  hot_x and the tables are 16-byte aligned, a tier is
    // padded to a multiple of 4.
    constexpr int HB = 4; // float4 per lane: tables of up to 16384 sources (one table, T == 1: up to 32768 in two rounds)
    const uint32_t Hpad = (H + 3u) & ~3u;
    f32x4 tnext[HB];
    auto tier_fetch = [&](uint32_t t, uint32_t round) {
        const uint32_t len = (Htot - t * H) < H ? (Htot - t * H) : H;
        const uint32_t H4 = (len + 3u) / 4u;
        const float *src = hot_x + (size_t)t * H;
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const uint32_t i = tid + (round * HB + k) * PB_ACC_BLOCK;
            if (i < H4)
                tnext[k] = *reinterpret_cast<const f32x4 *>(src + 4u * i);
        }
    };
    auto tier_store = [&](uint32_t t, uint32_t round, float *table) {
        const uint32_t len = (Htot - t * H) < H ? (Htot - t * H) : H;
        const uint32_t H4 = (len + 3u) / 4u;
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const uint32_t i = tid + (round * HB + k) * PB_ACC_BLOCK;
            if (i < H4)
                *reinterpret_cast<f32x4 *>(table + 4u * i) = tnext[k];
        }
    };
    // the first tier with edges for this item at or behind t (T: none); uniform over the workgroup
    auto next_tier = [&](uint32_t t) {
        for (; H && t < T && ABL != 4 && ABL != 6; ++t) {
            uint32_t ha, hb;
            tier_range(t, ha, hb);
            if (hb > ha)
                break;
        }
        return (H && ABL != 4 && ABL != 6) ? t : T;
    };
    uint32_t tier = next_tier(0);
    float *table = hot; // the table the current tier's edges gather from
    if (tier < T)
        for (uint32_t round = 0; round * HB * PB_ACC_BLOCK * 4u < H; ++round) {
            tier_fetch(tier, round);
            tier_store(tier, round, table);
        }
    __syncthreads();
    for (uint32_t q0 = q_first; q0 < qe; q0 += STEP * D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (BF) {
                atomicAdd(&acc[d[k].a < Racc ? d[k].a : Racc], pb_to_fix(v[k].x));
                atomicAdd(&acc[d[k].b < Racc ? d[k].b : Racc], pb_to_fix(v[k].y));
                atomicAdd(&acc[d[k].c < Racc ? d[k].c : Racc], pb_to_fix(v[k].z));
                atomicAdd(&acc[d[k].d < Racc ? d[k].d : Racc], pb_to_fix(v[k].w));
            } else {
                if (d[k].a != PB_NULL)
                    atomicAdd(&acc[d[k].a], pb_to_fix(v[k].x));
                if (d[k].b != PB_NULL)
                    atomicAdd(&acc[d[k].b], pb_to_fix(v[k].y));
                if (d[k].c != PB_NULL)
                    atomicAdd(&acc[d[k].c], pb_to_fix(v[k].z));
                if (d[k].d != PB_NULL)
                    atomicAdd(&acc[d[k].d], pb_to_fix(v[k].w));
            }
            fetch1(q0 + (uint32_t)(k + D) * STEP, v[k], d[k]);
        }
    }
    // hot edges: 4 bytes each (row_in_bin << 16 | index inside the tier), the value comes from the tier's LDS table.
    // The (tier, batch) pairs form ONE software-pipelined sequence: the entries of the next batch — the first one of
    // the next tier, when this tier is done — are requested before this batch is processed, and the next tier's
    // table is requested at this tier's first batch.  A tier with a few thousand edges per bin is a single, partly
    // filled batch: without the pipeline across the tier switch every tier cost one exposed HBM round trip.
    if (H16 && tier < T) {
        // 2-byte records: eight per lane and batch in ONE 16-byte load (half the bytes of the 4-byte records); a record's
        // table index = the base index of the lane's eight records (2 more bytes per lane) + the deltas up to it.  Same
        // software pipeline as below.
        constexpr uint32_t BATCH = PB_ACC_BLOCK * 8u;
        const uint16_t *h16 = reinterpret_cast<const uint16_t *>(hot_ent);
        const uint32_t filler = (Racc << 2) | (Racc << 18);
        uint32_t base, h_end;
        tier_range(tier, base, h_end);
        auto fetch16 = [&](uint32_t b0, uint32_t end, uint4 &ee, uint32_t &bi) {
            const uint32_t h = b0 + tid * 8u;
            ee = make_uint4(filler, filler, filler, filler);
            bi = 0u;
            if (h < end) {
                ee = *reinterpret_cast<const uint4 *>(h16 + h);
                bi = hot_base[h / PB_H16_LANE];
            }
        };
        uint4 e, en;
        uint32_t bi, bin_ = 0u;
        fetch16(base, h_end, e, bi);
        uint32_t after = next_tier(tier + 1);
        if (after < T)
            tier_fetch(after, 0);
        for (;;) {
            uint32_t nbase = base + BATCH, nend = h_end, ntier = tier;
            if (nbase >= h_end) {
                ntier = after;
                if (ntier < T)
                    tier_range(ntier, nbase, nend);
            }
            if (ntier < T)
                fetch16(nbase, nend, en, bin_);
            {
                uint32_t idx = bi; // the lane's records one after the other: index, then the add
                auto two = [&](uint32_t w) {
                    idx += w & 3u;
                    if (((w >> 2) & 0x3FFFu) != Racc)
                        atomicAdd(&acc[(w >> 2) & 0x3FFFu], pb_to_fix(table[idx]));
                    idx += (w >> 16) & 3u;
                    if ((w >> 18) != Racc)
                        atomicAdd(&acc[w >> 18], pb_to_fix(table[idx]));
                };
                two(e.x), two(e.y), two(e.z), two(e.w);
            }
            if (ntier >= T)
                break;
            if (ntier != tier) {
                float *other = table == hot ? hot + Hpad : hot;
                tier_store(ntier, 0, other);
                __syncthreads();
                table = other;
                after = next_tier(ntier + 1);
                if (after < T)
                    tier_fetch(after, 0);
            }
            e = en, bi = bin_;
            tier = ntier, base = nbase, h_end = nend;
        }
    } else if (tier < T) {
        constexpr int HU = 2; // uint4 of hot records per lane and batch (1 / 2 / 3 measured alike: 2.70-2.72 ms per sweep, tools/runs/r04_call28.sh)
        constexpr uint32_t BATCH = PB_ACC_BLOCK * PB_VEC * HU; // entries of one batch
        uint32_t base, h_end; // this batch starts at `base` (workgroup-uniform) of the tier's range [.., h_end)
        tier_range(tier, base, h_end);
        auto fetch_hot = [&](uint32_t b0, uint32_t end, uint4(&ee)[HU]) {
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                const uint32_t h = b0 + tid * PB_VEC + (uint32_t)k * PB_ACC_BLOCK * PB_VEC;
                ee[k] = h < end ? *reinterpret_cast<const uint4 *>(hot_ent + h)
                                : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            }
        };
        uint4 e[HU], en[HU];
        fetch_hot(base, h_end, e);
        uint32_t after = next_tier(tier + 1); // the tier behind this one (T: none)
        if (after < T)
            tier_fetch(after, 0); // T > 1: H <= 16384, one round
        for (;;) {
            // where the next batch lies
            uint32_t nbase = base + BATCH, nend = h_end, ntier = tier;
            if (nbase >= h_end) {
                ntier = after;
                if (ntier < T)
                    tier_range(ntier, nbase, nend);
            }
            if (ntier < T)
                fetch_hot(nbase, nend, en);
#pragma unroll
            for (int k = 0; k < HU; ++k) {
                if (BF) { // a padding record (0xFFFFFFFF): the padding accumulator, any entry of the table
                    const uint32_t hl = Hpad - 1u;
                    atomicAdd(&acc[(e[k].x >> 16) < Racc ? (e[k].x >> 16) : Racc], pb_to_fix(table[(e[k].x & 0xFFFFu) < hl ? (e[k].x & 0xFFFFu) : hl]));
                    atomicAdd(&acc[(e[k].y >> 16) < Racc ? (e[k].y >> 16) : Racc], pb_to_fix(table[(e[k].y & 0xFFFFu) < hl ? (e[k].y & 0xFFFFu) : hl]));
                    atomicAdd(&acc[(e[k].z >> 16) < Racc ? (e[k].z >> 16) : Racc], pb_to_fix(table[(e[k].z & 0xFFFFu) < hl ? (e[k].z & 0xFFFFu) : hl]));
                    atomicAdd(&acc[(e[k].w >> 16) < Racc ? (e[k].w >> 16) : Racc], pb_to_fix(table[(e[k].w & 0xFFFFu) < hl ? (e[k].w & 0xFFFFu) : hl]));
                } else {
                    if (e[k].x != 0xFFFFFFFFu)
                        atomicAdd(&acc[e[k].x >> 16], pb_to_fix(table[e[k].x & 0xFFFFu]));
                    if (e[k].y != 0xFFFFFFFFu)
                        atomicAdd(&acc[e[k].y >> 16], pb_to_fix(table[e[k].y & 0xFFFFu]));
                    if (e[k].z != 0xFFFFFFFFu)
                        atomicAdd(&acc[e[k].z >> 16], pb_to_fix(table[e[k].z & 0xFFFFu]));
                    if (e[k].w != 0xFFFFFFFFu)
                        atomicAdd(&acc[e[k].w >> 16], pb_to_fix(table[e[k].w & 0xFFFFu]));
                }
            }
            if (ntier >= T)
                break;
            if (ntier != tier) {
                // the other buffer: every wavefront left it at the last barrier (it held the tier before this one)
                float *other = table == hot ? hot + Hpad : hot;
                tier_store(ntier, 0, other);
                __syncthreads();
                table = other;
                after = next_tier(ntier + 1);
                if (after < T)
                    tier_fetch(after, 0);
            }
#pragma unroll
            for (int k = 0; k < HU; ++k)
                e[k] = en[k];
            tier = ntier, base = nbase, h_end = nend;
        }
    }
    __syncthreads();
    if (item.nparts > 1) {
        // An over-long bin is accumulated by several workgroups; integer partial sums commute, so the
        // last one to arrive adds them up and runs the epilogue.  Hand-off per the agent-scope
        // release/acquire recipe (cdna_hip_programming.md, Guideline 16).
        unsigned long long *mine = partials + (size_t)(item.slot0 + item.part) * Racc;
        for (uint32_t i = tid; i < Racc; i += PB_ACC_BLOCK)
            mine[i] = acc[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t prev = atomicAdd(&tickets[b], 1u);
            is_last = prev == item.nparts - 1u;
            if (is_last) {
                st_agent(&tickets[b], 0u); // ready for the next sweep
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (!is_last) {
            pb_err_fold(fold); // (it owns no error slot; it is one of the launch's workgroups)
            return;
        }
    }
    double err = 0.0;
    const uint32_t r0 = b * R;
    if (ABL == 3) {
        if (acc[tid % Racc] == 0x123456789ull)
            bin_err[b] = 1.0;
        return;
    }
    const bool vec_ok = (((uintptr_t)scores | (uintptr_t)x_out | (uintptr_t)outdeg) & 15u) == 0;
    if (item.nparts == 1 && (R & 3u) == 0 && vec_ok) {
        // Fused epilogue, 4 consecutive rows per lane and PB_EPI groups in flight: with one workgroup
        // per CU nothing else hides this phase's load latency (measured: 0.56 ms of a 1.7 ms kernel
        // at scale 26 when it ran one row per lane and iteration).
        constexpr int E = PB_EPI;
        for (uint32_t i0 = tid * 4u; i0 < R; i0 += PB_ACC_BLOCK * 4u * E) {
            U16x4 c4[E];
            f32x4 old4[E];
            uint4 od4[E];
            bool full[E];
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const uint32_t i = i0 + (uint32_t)k * PB_ACC_BLOCK * 4u;
                const uint32_t r = r0 + i;
                full[k] = i < R && r + 3u < n_local;
                if (full[k]) {
                    const u32x2 raw = *reinterpret_cast<const u32x2 *>(cidx + r);
                    c4[k] = U16x4{(uint16_t)raw.x, (uint16_t)(raw.x >> 16), (uint16_t)raw.y, (uint16_t)(raw.y >> 16)};
                    // a hub row among the four is finished by the hub kernels, possibly right now: no vector store
                    full[k] = c4[k].a != PB_HUBROW && c4[k].b != PB_HUBROW && c4[k].c != PB_HUBROW && c4[k].d != PB_HUBROW;
                }
                if (full[k]) {
                    old4[k] = *reinterpret_cast<const f32x4 *>(scores + r);
                    od4[k] = *reinterpret_cast<const uint4 *>(outdeg + r);
                }
            }
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const uint32_t i = i0 + (uint32_t)k * PB_ACC_BLOCK * 4u;
                const uint32_t r = r0 + i;
                if (full[k]) {
                    const uint16_t cs4[4] = {c4[k].a, c4[k].b, c4[k].c, c4[k].d};
                    const float olds[4] = {old4[k].x, old4[k].y, old4[k].z, old4[k].w};
                    const uint32_t ods[4] = {od4[k].x, od4[k].y, od4[k].z, od4[k].w};
                    float nw[4], xo[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned long long sum = cs4[j] != PB_NULL ? acc[cs4[j]] : 0ull;
                        const float incoming = (float)sum * PB_FIX_INV; // one rounding: the exactly rounded row sum
                        nw[j] = pr_new_score(base, damping, incoming);
                        xo[j] = __fdiv_rn(nw[j], (float)ods[j]);
                        err += fabs((double)__fsub_rn(nw[j], olds[j]));
                    }
                    f32x4 o;
                    o.x = nw[0], o.y = nw[1], o.z = nw[2], o.w = nw[3];
                    *reinterpret_cast<f32x4 *>(scores + r) = o;
                    o.x = xo[0], o.y = xo[1], o.z = xo[2], o.w = xo[3];
                    *reinterpret_cast<f32x4 *>(x_out + r) = o;
                } else if (i < R) { // the last rows of the slice, or four rows with a hub row among them
                    for (uint32_t rr = r; rr < n_local && rr < r + 4u; ++rr) {
                        const uint32_t c = cidx[rr];
                        if (c == PB_HUBROW)
                            continue;
                        const unsigned long long sum = c != PB_NULL ? acc[c] : 0ull;
                        err += pr_finalize(rr, (float)sum * PB_FIX_INV, base, damping, outdeg, scores, x_out);
                    }
                }
            }
        }
    } else {
        for (uint32_t i = tid; i < R; i += PB_ACC_BLOCK) {
            const uint32_t r = r0 + i;
            if (r < n_local) {
                const uint32_t c = cidx[r]; // rows without in-edges own no accumulator: incoming = 0
                if (c == PB_HUBROW)
                    continue; // finished by the hub kernels
                unsigned long long sum = 0ull;
                if (c != PB_NULL) {
                    if (item.nparts > 1) { // slices of one bin own consecutive partial slots [slot0, slot0 + nparts)
                        for (uint32_t k = 0; k < item.nparts; ++k)
                            sum += partials[(size_t)(item.slot0 + k) * Racc + c];
                    } else {
                        sum = acc[c];
                    }
                }
                const float incoming = (float)sum * PB_FIX_INV; // one rounding: the exactly rounded row sum
                err += pr_finalize(r, incoming, base, damping, outdeg, scores, x_out);
            }
        }
    }
    const double total = block_sum<double, PB_ACC_BLOCK / kWave>(err, red);
    if (tid == 0)
        st_agent(&bin_err[b], total);
    pb_err_fold(fold);
}

// ---- hub rows: the reference's own sums ---------------------------------------------------------------------------------
// The reference adds a row's terms left to right in f32 (page_rank.rs:143-146).  Rounds 2 and 3 imitated that with integer
// counts of ulps per 4096-entry step (pb_hub_kernel: ~300 wavefront instructions per step on sixteen wavefronts — measured
// in round 4, profiles/r04_accum_corun_ab.txt: as many vector instructions as the accumulate kernel itself for 22 % of the
// edges).  Here the sum is simply COMPUTED that way.  Rows below hub_long terms: a group's stream is sorted by source with
// its rows interleaved, so at plan time every 2048-entry block gets a stable permutation that makes it row-major
// (pb_hubseq_layout_kernel: p2_dst holds the entry's place in the block's LDS arrangement, `rows` the first place and the
// number of terms of every row), and per block the workgroup's four wavefronts scatter the values into LDS while lane g
// of wavefront 0 adds row g's terms in order: S = S + v, one v_add_f32 per term.  The sum is the reference's bit for bit
// for the same out_scores, whatever the partition.  What stays serial is the chain of a row's adds (~5 cycles per term):
// longer rows go to pb_hublong_kernel below.
// per group of pb_hubseq_kernel (hub_items order): where its stretch of hubsrc / of the sorted hot-hub keys / of hh_ent begins
struct PbSeqGroup {
    uint32_t q0, q1;     // its part of the value stream
    uint32_t hk0, hk1;   // its hot-hub keys (sorted by source)
    uint32_t ent0;       // its first hot record
    uint32_t blk0, nblk; // its blocks
    uint32_t nh;         // rows
};

// Block b of a group = the entries [b CAP, (b + 1) CAP) of the group's terms in source order, cold ones (positions of the value
// stream, padding entries counted: they read as the source before them) and hot ones merged — a merge-path split per block
// boundary: how many of the first R entries are cold.  A source is hot or cold for all its edges, so no two entries of the two
// lists compare equal.
__device__ __forceinline__ uint32_t pb_seq_split(const uint32_t *__restrict__ csrc, uint32_t nc, const uint64_t *__restrict__ hk,
                                                 uint32_t nhot, uint64_t smask, uint32_t R)
{
    uint32_t lo = R > nhot ? R - nhot : 0u, hi = R < nc ? R : nc;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1); // candidates: mid cold entries, R - mid hot ones
        // cold[mid] belongs to the first R entries iff it lies before hot[R - mid - 1]
        if (csrc[mid] < (uint32_t)(hk[R - mid - 1u] & smask))
            lo = mid + 1u;
        else
            hi = mid;
    }
    return lo;
}

__global__ void pb_hubseq_blocks_kernel(const PbSeqGroup *__restrict__ groups, uint32_t n_groups, uint32_t n_blocks,
                                        const uint32_t *__restrict__ hubsrc, uint32_t hub_q0, const uint64_t *__restrict__ hk,
                                        uint64_t smask, uint4 *__restrict__ blk)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks)
        return;
    const uint32_t g = (uint32_t)lower_bound_fn(0, n_groups, (uint64_t)b + 1, [&](uint64_t k) { return (uint64_t)groups[k].blk0 + groups[k].nblk; });
    const PbSeqGroup gr = groups[g];
    const uint32_t nc = gr.q1 - gr.q0, nhot = gr.hk1 - gr.hk0, total = nc + nhot;
    const uint32_t *cs = hubsrc + (gr.q0 - hub_q0);
    const uint64_t *hs = hk + gr.hk0;
    const uint32_t k = b - gr.blk0;
    const uint32_t R0 = k * PB_SEQ_CAP < total ? k * PB_SEQ_CAP : total, R1 = (k + 1u) * PB_SEQ_CAP < total ? (k + 1u) * PB_SEQ_CAP : total;
    const uint32_t i0 = pb_seq_split(cs, nc, hs, nhot, smask, R0), i1 = pb_seq_split(cs, nc, hs, nhot, smask, R1);
    blk[b] = make_uint4(gr.q0 + i0, gr.q0 + i1, gr.ent0 + (R0 - i0), gr.ent0 + (R1 - i1));
}

// One workgroup per block: the places of the block's terms in its row-major LDS arrangement.  Merged index of a cold entry =
// its index among the cold ones + the hot sources before it (a search in LDS), of a hot one likewise; then the stable rank of
// every entry among its row's entries in merged order (7-bit ballots per wavefront, wavefronts through a histogram), rows
// padded to 16 floats.  p2_dst: the place of every cold entry (PB_NULL: padding); hh_ent: place << 18 | hot rank.
__global__ __launch_bounds__(PB_SEQ_STEP / PB_VEC) void pb_hubseq_layout_kernel(const PbSeqGroup *__restrict__ groups, uint32_t n_groups,
                                                                               const uint4 *__restrict__ blk,
                                                                               const uint32_t *__restrict__ hubsrc, uint32_t hub_q0,
                                                                               const uint64_t *__restrict__ hk, uint64_t smask, int slot_shift,
                                                                               uint32_t slot_mask, const uint32_t *__restrict__ hot_rank,
                                                                               uint16_t *__restrict__ p2_dst, uint32_t *__restrict__ hh_ent,
                                                                               uint32_t *__restrict__ rows)
{
    constexpr uint32_t STEP = PB_SEQ_STEP, THREADS = STEP / PB_VEC, NWV = THREADS / kWave, NONE = PB_HUB_MAX;
    __shared__ uint32_t csrc[STEP], hsrc[STEP];
    __shared__ uint8_t rowof[STEP];
    __shared__ uint16_t placeof[STEP];
    __shared__ uint32_t hist[NWV][PB_HUB_MAX + 1]; // terms of every row per wavefront, then their exclusive prefix
    __shared__ uint32_t rbase[PB_HUB_MAX + 1];
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const uint32_t g = (uint32_t)lower_bound_fn(0, n_groups, (uint64_t)blockIdx.x + 1, [&](uint64_t k) { return (uint64_t)groups[k].blk0 + groups[k].nblk; });
    const PbSeqGroup gr = groups[g];
    const uint4 bk = blk[blockIdx.x];
    const uint32_t nc = bk.y - bk.x, nhot = bk.w - bk.z; // nc + nhot <= PB_SEQ_CAP
    const uint64_t *hs = hk + gr.hk0 + (bk.z - gr.ent0);
    for (uint32_t i = tid; i < NWV * (PB_HUB_MAX + 1); i += THREADS)
        (&hist[0][0])[i] = 0u;
    for (uint32_t i = tid; i < STEP; i += THREADS) {
        csrc[i] = i < nc ? hubsrc[bk.x - hub_q0 + i] : 0xFFFFFFFFu;
        hsrc[i] = i < nhot ? (uint32_t)(hs[i] & smask) : 0xFFFFFFFFu;
        rowof[i] = (uint8_t)NONE;
    }
    __syncthreads();
    // merged index and row of my entries: cold 4 t .. 4 t + 3, hot 4 t .. 4 t + 3
    uint32_t cmi[4], hmi[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t pc = tid * 4u + k;
        cmi[k] = 0xFFFFFFFFu, hmi[k] = 0xFFFFFFFFu;
        if (pc < nc) {
            const uint32_t slot = p2_dst[bk.x + pc];
            if (slot < gr.nh) { // a real entry (padding: PB_NULL)
                const uint32_t before = (uint32_t)lower_bound_fn(0, nhot, (uint64_t)csrc[pc], [&](uint64_t j) { return (uint64_t)hsrc[j]; });
                cmi[k] = pc + before;
                rowof[cmi[k]] = (uint8_t)slot;
            }
        }
        if (pc < nhot) {
            // cold positions (padding included: it reads as the source before it) in front of this hot source
            const uint32_t before = (uint32_t)lower_bound_fn(0, nc, (uint64_t)hsrc[pc], [&](uint64_t j) { return (uint64_t)csrc[j]; });
            hmi[k] = pc + before;
            rowof[hmi[k]] = (uint8_t)((uint32_t)(hs[pc] >> slot_shift) & slot_mask);
        }
    }
    __syncthreads();
    // stable rank among the entries of the same row, in merged order
    uint32_t s4[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        s4[k] = rowof[tid * 4u + k];
    uint64_t bm[4][7];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int bit = 0; bit < 7; ++bit)
            bm[k][bit] = __ballot((s4[k] >> bit) & 1u);
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t c = 0;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            uint64_t m = ~0ull;
#pragma unroll
            for (int bit = 0; bit < 7; ++bit)
                m &= ((s4[k] >> bit) & 1u) ? bm[k2][bit] : ~bm[k2][bit];
            c += (uint32_t)__popcll(m & lt) + ((k2 < k && s4[k2] == s4[k]) ? 1u : 0u);
        }
        rank[k] = c;
        if (s4[k] != NONE)
            atomicAdd(&hist[wave][s4[k]], 1u);
    }
    __syncthreads();
    if (tid < PB_HUB_MAX) {
        uint32_t c = 0;
        for (uint32_t w = 0; w < NWV; ++w) {
            const uint32_t t = hist[w][tid];
            hist[w][tid] = c;
            c += t;
        }
        rbase[tid] = c;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t at = 0;
        for (uint32_t r = 0; r < PB_HUB_MAX; ++r) {
            const uint32_t c = rbase[r];
            rbase[r] = at;
            rows[(size_t)blockIdx.x * PB_HUB_MAX + r] = (at << 16) | c;
            at += (c + PB_SEQ_PAD - 1u) & ~(PB_SEQ_PAD - 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (s4[k] != NONE)
            placeof[tid * 4u + k] = (uint16_t)(rbase[s4[k]] + hist[wave][s4[k]] + rank[k]);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t pc = tid * 4u + k;
        if (cmi[k] != 0xFFFFFFFFu)
            p2_dst[bk.x + pc] = placeof[cmi[k]];
        if (hmi[k] != 0xFFFFFFFFu)
            hh_ent[bk.z + pc] = ((uint32_t)placeof[hmi[k]] << 18) | hot_rank[hsrc[pc]];
    }
}

// One hub group per workgroup; lane g of wavefront 0 owns row g and adds its terms in CSR order, the other wavefronts only
// move data.  A round = one block of 2040 terms: the walk of block b, then block b + 1 is scattered into the buffer.
//
// The memory pipeline is written so that NO wait in the loop is a wait for everything.  The vector-memory counter completes
// in order: waiting for a load waits for every older one, and the compiler can leave younger ones in flight only if it can
// COUNT them — so every load of a round is unconditional (indices clamped instead of branches; what lies outside a block's
// ranges is dropped when it is used) and the rounds issue in a fixed order:
//     G(b+2)  values of block b + 2's hot terms from hot_x (L2)        needs R(b+2), the oldest load of the round before
//     R(b+3)  hot records of block b + 3
//     V(b+2)  values + places of block b + 2's stretch of the stream
//     I(b+2)  its row table
//     walk b, barrier, scatter b + 1 (needs G, V, I of b + 1: issued one round ago, 13 younger loads stay in flight), barrier
// Until round 4's last day the loads sat behind `if (q < end)`: the compiler then has to assume that nothing younger was
// issued and waits with vmcnt(0) — for the block requested a moment ago as well.  Every round paid a full memory latency
// under the accumulate kernel's traffic (~8 us per block; the walk of a block's 32..120 terms per row is 0.2..0.8 us).
// (at most 96 VGPRs: one wavefront of this kernel, two of pb_hublong_kernel's and four of the accumulate kernel's share a SIMD)
//
// WG threads (round 6): what a round costs is mostly the instructions of ONE wavefront per SIMD — with the walk, the scatter, the hot
// addresses and the padding zeros taken out one by one (a throw-away build, tools/runs/r06_call31.sh) a launch of the block-Gauss-
// Seidel call went 78 -> 46 us at scale 22 and 414 -> 237 at scale 26 without the walk and stayed there without the rest: 1.4-1.7 us
// of a 2.4-3.2 us round are neither memory latency nor the adds.  (Loading a block's ranges once instead of three dependent scalar
// loads per round, and skewing the rows' LDS stretches over the 16-byte slots, changed nothing: r06_call32.sh.)  Twice the threads
// halve every wavefront's share of the staging: 66 / 358 us per launch (r06_call32.sh, second pass).  512 threads are what a launch
// gets whose walks ARE the critical path — a part's hub rows in a sweep in blocks, a partition slice's; beside the accumulate kernel
// of a whole synchronous sweep, where the walks have slack, their 2 x 80 registers per SIMD crowd the other two kernels (scale 26:
// pb_hubseq_kernel 713 -> 968 us, pb_hublong_kernel 674 -> 921, the sweep +2-3 %): those keep 256.
template <uint32_t WG>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 8))) void pb_hubseq_kernel(const float *__restrict__ vals, const uint16_t *__restrict__ p2_dst,
                                                              const PbHubItem *__restrict__ items,
                                                              const uint32_t *__restrict__ blk_first,
                                                              const uint4 *__restrict__ blk, const uint32_t *__restrict__ rows,
                                                              const uint32_t *__restrict__ hh_ent, const float *__restrict__ hot_x,
                                                              const uint32_t *__restrict__ hub_rows,
                                                              const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                                              float *__restrict__ x_out, double *__restrict__ group_err, float base,
                                                              float damping, uint32_t v_safe, uint32_t h_safe, uint32_t n_groups, uint32_t walk_prio,
                                                              PbErrFold fold, const uint32_t *__restrict__ grp_list)
{
    constexpr uint32_t STEP = PB_SEQ_STEP;                      // stream entries one round of loads covers
    constexpr int PER = (int)(STEP / (WG * PB_VEC));     // float4 + 4 places per thread and block
    constexpr int HP = (int)(PB_SEQ_HOT / WG);                        // hot records per thread and block in the pipeline
    constexpr uint32_t DUMP = PB_SEQ_BUF;                       // where everything that is not a term lands
    __shared__ __attribute__((aligned(16))) float buf[PB_SEQ_BUF + 4];
    __shared__ double red[WG / kWave];
    const uint32_t tid = threadIdx.x;
    if (walk_prio && tid < kWave)
        __builtin_amdgcn_s_setprio(3); // the walk is the group's critical path: first in line at its SIMD's issue
    // the grid may be smaller than the number of groups (pb_hub_dispatch): workgroup w then takes groups w, w + grid, ...
    // grp_list (a sweep in row blocks, pb_set_parts with hub_by_part): the groups of this launch, n_groups of them
    for (uint32_t gi = blockIdx.x; gi < n_groups; gi += gridDim.x) {
    const uint32_t grp = grp_list ? grp_list[gi] : gi;
    const PbHubItem item = items[grp]; // longest groups first
    const uint32_t nh = item.nh;
    const bool walker = tid < nh; // nh <= 64: lane g of wavefront 0 owns row g
    const uint32_t b0 = blk_first[grp], nb = blk_first[grp + 1] - b0;
    const uint32_t *ri = rows + (size_t)b0 * PB_HUB_MAX + (tid & (PB_HUB_MAX - 1u));
    const uint4 *bt = blk + b0;
    // block k's ranges {stream first, stream end, hot records first, hot records end}: the same for every lane (scalar loads);
    // beyond the group's last block: the last block again (loaded, never used)
    const uint32_t last = nb ? nb - 1u : 0u;
    auto range_of = [&](uint32_t k) { return bt[k < last ? k : last]; };
    struct Stream { // V(k), raw
        f32x4 v[PER];
        u32x2 d[PER];
    } sa, sb;
    struct Hot { // G(k), raw, and where the values go: two 12-bit places per register (DUMP for what is no record)
        float x[HP];
        uint32_t at[HP / 2];
    } ha, hb;
    uint32_t rx[HP]; // R(k), raw: place << 18 | hot rank
    auto issue_r = [&](uint32_t k) {
        const uint4 rg = range_of(k);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const uint32_t h = rg.z + (uint32_t)j * WG + tid;
            rx[j] = hh_ent[h < h_safe ? h : h_safe];
        }
    };
    auto issue_g = [&](uint32_t k, Hot &ht) { // needs R(k) in rx
        const uint4 rg = range_of(k);
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const bool is = rg.z + (uint32_t)j * WG + tid < rg.w;
            ht.x[j] = hot_x[is ? rx[j] & 0x3FFFFu : 0u];
            const uint32_t at = is ? rx[j] >> 18 : DUMP;
            if (j & 1)
                ht.at[j / 2] |= at << 16;
            else
                ht.at[j / 2] = at;
        }
    };
    auto issue_v = [&](uint32_t k, Stream &st) {
        const uint32_t qa = range_of(k).x & ~3u;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t q = qa + ((uint32_t)j * WG + tid) * PB_VEC, qc = q < v_safe ? q : v_safe;
            st.v[j] = *reinterpret_cast<const f32x4 *>(vals + qc);
            st.d[j] = *reinterpret_cast<const u32x2 *>(p2_dst + qc);
        }
    };
    auto issue_i = [&](uint32_t k) { return ri[(size_t)(k < last ? k : last) * PB_HUB_MAX]; };
    auto scatter = [&](uint32_t k, const Stream &st, const Hot &ht) { // no branch per term: what is not a term goes to DUMP
        const uint4 rg = range_of(k);
        const uint32_t qa = rg.x & ~3u;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t q = qa + ((uint32_t)j * WG + tid) * PB_VEC;
            const uint32_t p0 = st.d[j].x & 0xFFFFu, p1 = st.d[j].x >> 16, p2 = st.d[j].y & 0xFFFFu, p3 = st.d[j].y >> 16;
            buf[(p0 < DUMP && q >= rg.x && q < rg.y) ? p0 : DUMP] = st.v[j].x;
            buf[(p1 < DUMP && q + 1u >= rg.x && q + 1u < rg.y) ? p1 : DUMP] = st.v[j].y;
            buf[(p2 < DUMP && q + 2u >= rg.x && q + 2u < rg.y) ? p2 : DUMP] = st.v[j].z;
            buf[(p3 < DUMP && q + 3u >= rg.x && q + 3u < rg.y) ? p3 : DUMP] = st.v[j].w;
        }
#pragma unroll
        for (int j = 0; j < HP; ++j)
            buf[(ht.at[j / 2] >> ((j & 1) * 16)) & 0xFFFFu] = ht.x[j];
        // a block with more than 1024 hot terms (rare: more than half of its terms): the rest without the pipeline
        for (uint32_t h = rg.z + (uint32_t)HP * WG + tid; h < rg.w; h += WG) {
            const uint32_t rec = hh_ent[h];
            buf[rec >> 18] = hot_x[rec & 0x3FFFFu];
        }
    };
    // zeros behind a row's terms up to its 16-float boundary: the walk adds whole steps (x + 0 = x)
    auto pads = [&](uint32_t info) {
        const uint32_t at = (info >> 16) + (info & 0xFFFFu), end = (info >> 16) + (((info & 0xFFFFu) + PB_SEQ_PAD - 1u) & ~(PB_SEQ_PAD - 1u));
#pragma unroll
        for (uint32_t j = 0; j < PB_SEQ_PAD - 1u; ++j)
            if (at + j < end)
                buf[at + j] = 0.0f;
    };
    // block 0 without the pipeline, then the pipeline's state at round 0: R(2) in flight, G(1), V(1), I(1) behind it
    uint32_t info, info_a = 0, info_b = 0; // row table of the block in the buffer / of the block in sa / sb
    issue_r(0);
    issue_v(0, sa);
    info = issue_i(0);
    issue_g(0, ha);
    if (walker)
        pads(info);
    scatter(0, sa, ha);
    issue_r(1u);
    issue_g(1u, ha);
    issue_r(2u);
    issue_v(1u, sa);
    info_a = issue_i(1u);
    lds_barrier();
    float S = 0.0f; // page_rank.rs:143: the row's sum starts at zero ...
    // round b: `cur` / `hcur` / `info_cur` hold block b + 1 (requested one round ago), block b + 2 goes into `nxt` / `hnxt`
    auto round = [&](uint32_t b, Stream &cur, Hot &hcur, uint32_t &info_cur, Stream &nxt, Hot &hnxt, uint32_t &info_nxt) {
        issue_g(b + 2u, hnxt);
        issue_r(b + 3u);
        issue_v(b + 2u, nxt);
        info_nxt = issue_i(b + 2u);
        const bool more = b + 1u < nb;
        if (walker) {
            uint32_t k = (info >> 16) / 4u;
            const uint32_t end = k + (((info & 0xFFFFu) + PB_SEQ_PAD - 1u) / PB_SEQ_PAD) * (PB_SEQ_PAD / 4u);
            if (k < end) {
                // Two register sets in turn, no copies: the next step's 16 terms are requested before this step's are added
                // (16 dependent v_add_f32).  Every term is added to the sum in CSR order, each add rounded to f32
                // (page_rank.rs:144-146).
                constexpr uint32_t Q = PB_SEQ_PAD / 4u;
                f32x4 a0, a1, a2, a3, n0, n1, n2, n3;
                // The LDS reads are INLINE ASSEMBLY with hand-placed waits (round 6).  Written as plain loads, the request for the
                // next step sat behind `if (k + Q < end)`: the compiler could not count the reads in flight and waited for the step
                // it had just requested before the first add of the one at hand (s_waitcnt lgkmcnt(3) with four younger reads
                // outstanding) — every step paid an LDS round trip; made unconditional, the compiler moved each request down to its
                // first use, same round trip.  Here the next step's four reads are requested (behind a row's last step they read
                // whatever follows, clamped to the buffer, and nothing is added), then lgkmcnt(4) — LDS returns in order: the step
                // at hand has arrived, the four younger reads stay in flight under its 16 dependent adds.  (Waits the compiler
                // places for its own LDS accesses stay correct with older or younger reads of these outstanding: a count-based wait
                // can only wait for more.)
                constexpr uint32_t K_SAFE = (PB_SEQ_BUF + 4u) / 4u - Q;
                const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(buf); // (a flat LDS address: its low half is the offset)
#define GM_SEQ_GET(x0, x1, x2, x3, at)                                                                                      \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48" \
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)                                                                \
                 : "v"(lds0 + (at) * 16u)                                                                                   \
                 : "memory")
#define GM_SEQ_ARRIVED(x0, x1, x2, x3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3))
#define GM_SEQ_ADD(x0, x1, x2, x3)                                                                                          \
    S = __fadd_rn(S, x0.x), S = __fadd_rn(S, x0.y), S = __fadd_rn(S, x0.z), S = __fadd_rn(S, x0.w);                         \
    S = __fadd_rn(S, x1.x), S = __fadd_rn(S, x1.y), S = __fadd_rn(S, x1.z), S = __fadd_rn(S, x1.w);                         \
    S = __fadd_rn(S, x2.x), S = __fadd_rn(S, x2.y), S = __fadd_rn(S, x2.z), S = __fadd_rn(S, x2.w);                         \
    S = __fadd_rn(S, x3.x), S = __fadd_rn(S, x3.y), S = __fadd_rn(S, x3.z), S = __fadd_rn(S, x3.w)
                GM_SEQ_GET(a0, a1, a2, a3, k);
                for (;;) {
                    GM_SEQ_GET(n0, n1, n2, n3, (k + Q < K_SAFE ? k + Q : K_SAFE));
                    GM_SEQ_ARRIVED(a0, a1, a2, a3);
                    GM_SEQ_ADD(a0, a1, a2, a3);
                    k += Q;
                    if (k >= end)
                        break;
                    GM_SEQ_GET(a0, a1, a2, a3, (k + Q < K_SAFE ? k + Q : K_SAFE));
                    GM_SEQ_ARRIVED(n0, n1, n2, n3);
                    GM_SEQ_ADD(n0, n1, n2, n3);
                    k += Q;
                    if (k >= end)
                        break;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef GM_SEQ_ADD
#undef GM_SEQ_ARRIVED
#undef GM_SEQ_GET
            }
            if (more)
                pads(info_cur); // the next block's arrangement: these slots are not among the places its terms are scattered to
        }
        lds_barrier(); // the walk is over: the buffer may be overwritten
        if (more)
            scatter(b + 1u, cur, hcur);
        info = info_cur;
        lds_barrier();
    };
    for (uint32_t b = 0; b < nb; b += 2u) {
        round(b, sa, ha, info_a, sb, hb, info_b);
        if (b + 1u < nb)
            round(b + 1u, sb, hb, info_b, sa, ha, info_a);
    }
    double err = 0.0;
    if (walker)
        err = pr_finalize(hub_rows[item.row0 + tid], S, base, damping, outdeg, scores, x_out);
    const double total = block_sum<double, WG / kWave>(err, red);
    if (tid == 0)
        st_agent(&group_err[item.group], total);
    lds_barrier();
    } // groups of this workgroup
    pb_err_fold(fold);
}

// ---- long rows: the same sums, in parallel ----------------------------------------------------------------------------
// S_{k+1} = fl(S_k + v_k) looks like a chain, but while S stays inside one binade [2^e, 2^(e+1)) it is integer arithmetic:
// with ulp = 2^(e-23), S = J ulp (2^23 <= J < 2^24) and v = t ulp, fl(S + v) = RNE(J + t) ulp, and RNE(J + t) = J + rint(t)
// unless t ends in exactly .5, where the result is the EVEN one of J + floor(t) and J + floor(t) + 1: it depends on J only
// through its parity.  So a run of terms acts on J as (count added when the run starts on an even J, count added when it
// starts on an odd J), and two runs compose into such a pair again — an associative operation, hence a scan:
//   every thread takes 16 consecutive terms and forms its pair (one pass over the terms when none of them is a tie);
//   an exclusive scan over the workgroup gives every thread the J its run starts from;
//   the first thread whose run ends at or beyond 2^24 — S leaves the binade there — starts from an exactly known S and
//   adds its 16 terms with v_add_f32, one after the other; everything behind it is redone on the new grid.
// S leaves a binade a few dozen times per row (and at every doubling of the first few thousand terms), so a PASS — a
// super-block of 8192 terms — costs one round (one barrier), sometimes two.  Bit for bit the reference's sum:
// tests/test_gpu_hub_adversarial.py compares rows of up to 2^20 + 1 terms with orc_page_rank_jacobi_sweep's sequential sums
// for equality.
//
// A ROW OVER SEVERAL WORKGROUPS (round 5).  One workgroup per row made the 854,315-term row of RMAT scale 26 a chain of 105
// passes (0.35 ms: the critical path of its rank in an 8-way partition).  A row longer than `passes_per_item` passes is now
// cut into ITEMS of that many passes, one workgroup each:
//   1. (no dependence on the items before) for every pass whose sum stayed inside ONE binade in the PREVIOUS sweep — the
//      sums at the pass boundaries are kept per engine (`sbs`); between two sweeps of a converging iteration they move by
//      ulps — the workgroup forms the pass's pair (count from an even J, count from an odd J) on that binade's grid;
//   2. it waits for the exact S the item before it hands over (a 64-bit word per item: launch epoch << 32 | bits of S;
//      items are drawn from a counter in row order, so the item waited for is always running or done);
//   3. pass by pass: if S lies in the binade the pair was formed for and J + count stays below 2^24, S = (J + count) ulp —
//      a few scalar operations; otherwise (a pass in which S leaves its binade, the first pass of a row, a prediction
//      that no longer holds, the first sweep of an engine) the pass is redone the way described above, from the exact S;
//   4. the exact S is handed on, or the row is finished (pr_finalize).
// What is predicted is only WHICH passes can be skipped through; every S is either (J + count) ulp with the count's grid
// verified against the exact S, or the result of the sequential pass — the bits cannot depend on the prediction.
// (at most 96 VGPRs: two of its wavefronts, one of pb_hubseq_kernel's (112) and four of the accumulate kernel's (56) share a SIMD's 512)
constexpr uint32_t PB_LONG_PMAX = 16;  // passes of an item, at most

// GATHER (a plan whose hub rows' lists are NOT ascending — CsrLayout::Unsorted, the reference's default, csr.rs:34-45 — so that the
// order the value stream delivers a row's terms in, ascending source, is not the order of page_rank.rs:143-146): EVERY hub row
// is an item list of this kernel, its range [q0, q1) counts entries of `gidx`, which names for the row's k-th in-neighbour IN
// CSR ORDER a position of the value stream that holds that source's out_score (any entry of the row's group with that source:
// a 1 MiB stretch, L2-resident while the group's rows are summed).  4 bytes and an L2 gather more per hub term than the
// source-ordered path; the sums are the reference's for whatever order the lists are in.
template <bool GATHER>
__global__ __launch_bounds__(PB_LONG_WG) __attribute__((amdgpu_waves_per_eu(5, 8))) void pb_hublong_kernel(const float *__restrict__ vals, const uint16_t *__restrict__ p2_dst,
                                                                const uint32_t *__restrict__ gidx,
                                                                const PbHubItem *__restrict__ items,
                                                                const PbLongItem *__restrict__ litems, uint32_t n_items,
                                                                unsigned long long *__restrict__ ticket,
                                                                unsigned long long *handoff, float *sbs,
                                                                uint32_t epoch,
                                                                const uint32_t *__restrict__ hub_rows,
                                                                const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                                                float *__restrict__ x_out, double *__restrict__ group_err, float base,
                                                                float damping, PbErrFold fold, const uint32_t *__restrict__ item_list)
{
    constexpr uint32_t NWV = PB_LONG_WG / kWave, PER = PB_LONG_PER, SUPER = PB_LONG_WG * PER, SAT = 1u << 30, NONE = 0xFFFFFFFFu;
    constexpr uint32_t WARM = 1024 / PER; // threads whose terms (the row's first 1024) are added one after the other, see below
    // per wavefront: its runs composed, the first thread whose run leaves the binade; two copies used in turn: a wavefront
    // that leaves a round through its single barrier may write the next round's totals while another still reads these
    __shared__ uint32_t w_a0s[2][NWV], w_a1s[2][NWV], w_firsts[2][NWV];
    __shared__ float s_bcast;
    __shared__ uint32_t s_item, s_ahead;
    __shared__ uint32_t sum_t0[PB_LONG_PMAX], sum_t1[PB_LONG_PMAX], sum_e[PB_LONG_PMAX]; // the passes' pairs and their binade (0: none)
    __shared__ uint32_t pred_e[PB_LONG_PMAX]; // the binade a pass is predicted to stay in (read once, by one thread)
    constexpr uint32_t TROW = PER + 4;          // floats between two threads' rows in the turning buffer (conflict-free 16-byte reads)
    constexpr uint32_t TOWN = PB_LONG_WG * 4 / PER; // threads that own the 2048 terms of one round
    __shared__ __attribute__((aligned(16))) float tbuf[TOWN * TROW]; // 2048 terms of the super-block at a time: 10 KiB
    static_assert(PB_LONG_WG == 512 && (PB_LONG_PER == 16 || PB_LONG_PER == 32) && WARM <= (uint32_t)kWave, "turning buffer / warm-up layout");
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    uint32_t flip = 0;
    // the items in row order from a counter: exactly n_items draws per launch, so the counter needs no reset
    // item_list (a sweep in row blocks): the n_items items of this launch, every row's in pass order — whoever an item waits
    // for stands in front of it in the list as well
    if (tid == 0) {
        const uint32_t draw = (uint32_t)(atomicAdd(ticket, 1ull) % n_items);
        s_item = item_list ? item_list[draw] : draw;
    }
    lds_barrier();
    const uint32_t me = s_item;
    const PbLongItem li = litems[me];
    const PbHubItem item = items[li.row]; // one row: the entries of its group's stretch whose slot is item.nh (padding entries: PB_NULL)
    const uint32_t my_slot = item.nh;
    // (counts from an even / odd start) of run F followed by run G; saturated: beyond the first run that leaves the binade
    // nothing is used
    auto compose = [&](uint32_t f0, uint32_t f1, uint32_t g0, uint32_t g1, uint32_t &h0, uint32_t &h1) {
        const uint32_t n0 = f0 + ((f0 & 1u) ? g1 : g0), n1 = f1 + ((f1 & 1u) ? g0 : g1);
        h0 = n0 < SAT ? n0 : SAT, h1 = n1 < SAT ? n1 : SAT;
    };
    // The super-block at sb as the memory system likes it: lane-interleaved float4s (a thread reading its own 32 consecutive
    // terms touches a cache line of its own with every load — 64 lines per wavefront instruction, measured ~4 us per
    // super-block and CU).  The loaded registers are NOT touched here — the values and their slots stay raw until `turn`
    // needs them, one pass later: masking the padding entries at once made the loads synchronous (measured with cycle
    // stamps: 49 % of a row's time was spent "issuing" the next super-block's loads).
    struct Raw {
        f32x4 x[PER / 4];
        u32x2 d[PER / 4];
    } raw;
    auto fetch = [&](uint32_t sb) {
#pragma unroll
        for (uint32_t k = 0; k < PER / 4; ++k) {
            const uint32_t q = sb + (k * PB_LONG_WG + tid) * 4u;
            raw.d[k].x = raw.d[k].y = 0xFFFFFFFFu; // behind the row: padding
            raw.x[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (GATHER) {
                if (q < item.q1) { // (a row's stretch of gidx starts on a multiple of 4 and is padded with valid positions)
                    const uint4 ix = *reinterpret_cast<const uint4 *>(gidx + q);
                    raw.x[k] = f32x4{vals[ix.x], vals[ix.y], vals[ix.z], vals[ix.w]};
                    raw.d[k].x = (q + 1u < item.q1 ? 0u : 0xFFFF0000u);
                    raw.d[k].y = (q + 2u < item.q1 ? 0u : 0x0000FFFFu) | (q + 3u < item.q1 ? 0u : 0xFFFF0000u);
                }
            } else if (q < item.q1) {
                raw.x[k] = *reinterpret_cast<const f32x4 *>(vals + q);
                raw.d[k] = *reinterpret_cast<const u32x2 *>(p2_dst + q);
            }
        }
    };
    // ... and turned, 2048 terms at a time through 10 KiB of LDS, into PER CONSECUTIVE terms per thread (padding entries —
    // slot PB_NULL instead of 0 — become 0): the threads that own the round's terms read their rows.  raw.x[r] = round r.
    auto turn = [&](float(&out)[PER]) {
#pragma unroll
        for (uint32_t r = 0; r < PER / 4; ++r) {
            const uint32_t i = tid * 4u; // place inside the round's 2048 terms
            f32x4 x = raw.x[r];
            const u32x2 d = raw.d[r];
            x.x = (d.x & 0xFFFFu) == my_slot ? x.x : 0.0f; // (another row's term, or padding: + 0 leaves every sum as it is)
            x.y = (d.x >> 16) == my_slot ? x.y : 0.0f;
            x.z = (d.y & 0xFFFFu) == my_slot ? x.z : 0.0f;
            x.w = (d.y >> 16) == my_slot ? x.w : 0.0f;
            *reinterpret_cast<f32x4 *>(tbuf + (i / PER) * TROW + (i % PER)) = x;
            lds_barrier();
            if (tid / TOWN == r) {
#pragma unroll
                for (uint32_t k = 0; k < PER / 4; ++k) {
                    const f32x4 y = *reinterpret_cast<const f32x4 *>(tbuf + (tid % TOWN) * TROW + 4 * k);
                    out[4 * k] = y.x, out[4 * k + 1] = y.y, out[4 * k + 2] = y.z, out[4 * k + 3] = y.w;
                }
            }
            lds_barrier();
        }
    };
    float S = 0.0f; // page_rank.rs:143
    float v[PER];
    // my run of v on the grid of binade e as (count from an even J, count from an odd J); a term of 2^24 ulps or more makes the
    // count reach 2^24 by itself, which is what marks the run as one in which S leaves the binade
    auto my_pair = [&](float iu, uint32_t &a0, uint32_t &a1, bool &tie) {
        uint32_t sum = 0;
        tie = false;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const float t = __builtin_fminf(v[j] * iu, 16777216.0f); // exact: a power-of-two scaling
            const float r = __builtin_rintf(t);
            sum += (uint32_t)r;
            tie |= __builtin_fabsf(t - r) == 0.5f; // t - r is exact
        }
        a0 = a1 = sum; // <= PER x 2^24 < SAT
        if (tie) { // the two-state walk: a tie goes to the even J
            uint32_t x0 = 0, x1 = 0, p0 = 0, p1 = 1;
#pragma unroll 4
            for (uint32_t j = 0; j < PER; ++j) {
                const float t = __builtin_fminf(v[j] * iu, 16777216.0f), fl = __builtin_floorf(t);
                const uint32_t c = (uint32_t)__builtin_rintf(t), f = (uint32_t)fl;
                const bool half = (t - fl) == 0.5f;
                const uint32_t c0 = half ? f + ((p0 + f) & 1u) : c, c1 = half ? f + ((p1 + f) & 1u) : c;
                x0 += c0, x1 += c1;
                p0 = (p0 + c0) & 1u, p1 = (p1 + c1) & 1u;
            }
            a0 = x0, a1 = x1;
        }
    };
    // One pass from an exactly known S: rounds of "everyone's pair, the workgroup's scan, the first run that leaves the
    // binade added the slow way" until every thread's terms are in.
    auto seq_pass = [&](bool row_start) {
        uint32_t done = 0; // threads below `done` have had their terms added
        if (row_start) {
            // The row's first terms: a sum that starts at zero doubles after 2, 4, 8, ... terms, and every doubling would
            // be a round of its own.  The first 64 threads' terms (1024) are simply added in order by wavefront 0, the sum
            // handed from lane to lane: ~100 cycles per thread instead of a round of the whole workgroup per doubling.
            if (wave == 0) {
                float s = 0.0f;
                for (uint32_t k = 0; k < WARM; ++k) {
                    float mine = s;
#pragma unroll
                    for (uint32_t j = 0; j < PER; ++j)
                        mine = __fadd_rn(mine, v[j]); // page_rank.rs:144-146
                    s = __shfl(mine, (int)k, kWave);
                }
                if (lane == 0)
                    s_bcast = s;
            }
            lds_barrier();
            S = s_bcast;
            done = WARM;
        }
        for (;;) {
            uint32_t *w_a0 = w_a0s[flip], *w_a1 = w_a1s[flip], *w_first = w_firsts[flip];
            flip ^= 1u;
            const uint32_t sbits = __float_as_uint(S), e = sbits >> 23; // S >= 0
            const bool binade = e >= 24u && e < 255u;                   // S is a normal number with a usable grid
            const uint32_t J0 = (sbits & 0x7FFFFFu) | 0x800000u;
            const float iu = __uint_as_float((277u - (binade ? e : 150u)) << 23); // 1 / ulp(S)
            const float ulp = __uint_as_float(((binade ? e : 150u) - 23u) << 23);
            uint32_t a0 = 0, a1 = 0;
            bool tie = false;
            if (tid >= done) {
                if (!binade)
                    a0 = a1 = SAT; // S is still zero (or tiny): the first thread adds its terms the slow way
                else
                    my_pair(iu, a0, a1, tie);
            }
            // the wavefront's runs composed.  Without a tie in the wavefront a run adds the same count from either parity and
            // composing is adding: a butterfly sum of one value; otherwise the inclusive scan of the pairs.
            const bool wave_tie = __ballot(tie) != 0ull;
            uint32_t i0 = a0, i1 = a1; // the scan, when it is made
            bool scanned = false;
            auto scan = [&]() {
#pragma unroll
                for (uint32_t o = 1; o < (uint32_t)kWave; o <<= 1) {
                    const uint32_t f0 = (uint32_t)__shfl_up((int)i0, o, kWave), f1 = (uint32_t)__shfl_up((int)i1, o, kWave);
                    if (lane >= o)
                        compose(f0, f1, i0, i1, i0, i1);
                }
                scanned = true;
            };
            uint32_t wt0, wt1;
            if (wave_tie) {
                scan();
                wt0 = (uint32_t)__shfl((int)i0, kWave - 1, kWave), wt1 = (uint32_t)__shfl((int)i1, kWave - 1, kWave);
            } else {
                uint32_t x = a0;
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) {
                    x += (uint32_t)__shfl_xor((int)x, o, kWave);
                    x = x < SAT ? x : SAT;
                }
                wt0 = wt1 = x;
            }
            if (lane == 0)
                w_a0[wave] = wt0, w_a1[wave] = wt1;
            lds_barrier();
            uint32_t b0 = 0, b1 = 0, t0 = 0, t1 = 0; // the wavefronts before this one; all of them
#pragma unroll
            for (uint32_t w = 0; w < NWV; ++w) {
                if (w == wave)
                    b0 = t0, b1 = t1;
                compose(t0, t1, w_a0[w], w_a1[w], t0, t1);
            }
            const uint32_t P0 = J0 & 1u;
            if (binade && J0 + (P0 ? t1 : t0) < (1u << 24)) { // every remaining run stayed inside the binade:
                S = (float)(J0 + (P0 ? t1 : t0)) * ulp;        // S = (J0 + count) ulp, exactly — the common round, one barrier
                break;
            }
            // S leaves the binade somewhere: the first thread in whose run it does
            if (!scanned)
                scan();
            uint32_t e0 = (uint32_t)__shfl_up((int)i0, 1, kWave), e1 = (uint32_t)__shfl_up((int)i1, 1, kWave);
            if (lane == 0)
                e0 = e1 = 0u;
            const uint32_t bw = P0 ? b1 : b0;             // added by the wavefronts before this one
            const uint32_t pw = (P0 + bw) & 1u;
            const uint32_t bl = pw ? e1 : e0;             // ... and by the lanes before this one
            const uint32_t before = (bw + bl) < SAT ? bw + bl : SAT;
            const uint32_t mine = ((pw + bl) & 1u) ? a1 : a0;
            const bool cross = tid >= done && (!binade || J0 + before + mine >= (1u << 24));
            const uint64_t cm = __ballot(cross);
            if (lane == 0)
                w_first[wave] = cm ? wave * kWave + (uint32_t)__ffsll((unsigned long long)cm) - 1u : NONE;
            lds_barrier();
            uint32_t first = NONE;
#pragma unroll
            for (uint32_t w = 0; w < NWV; ++w)
                first = w_first[w] < first ? w_first[w] : first;
            if (tid == first) {
                float s = binade ? (float)(J0 + before) * ulp : S; // exact: J0 + before < 2^24
#pragma unroll
                for (uint32_t j = 0; j < PER; ++j)
                    s = __fadd_rn(s, v[j]); // page_rank.rs:144-146
                s_bcast = s;
            }
            lds_barrier();
            S = s_bcast;
            done = first + 1u;
            if (done >= PB_LONG_WG)
                break;
        }
    };
    const uint32_t q_first = item.q0 + li.pass0 * SUPER;
    float *my_sbs = sbs + li.sb0 + li.pass0; // the sums at my passes' boundaries: [p] before pass p, [npass] behind the last
    uint32_t held = NONE; // the pass whose entries are in `raw`
    // 1. the pairs of the passes that stayed inside one binade in the sweep before — only where there is something to wait for.
    // ONE thread reads the boundary sums, once: the items before and behind this one write the two outer ones (their exact S
    // of THIS sweep) while this one runs, so two threads could see two different sums — with different exponents when a sum
    // has just crossed a power of two — and disagree about which passes are formed ahead, i.e. about the barriers they meet,
    // or form their pairs on two different grids.  (The first version of this kernel had every thread read them for itself; no
    // wrong sum was ever traced to it — tools/debug_unsorted.py compares every hub row with the REFORDER engine sweep by sweep —
    // but nothing ruled it out either.)
    if (tid < PB_LONG_PMAX)
        sum_e[tid] = 0u;
    if (tid == 0) {
        uint32_t mask = 0;
        if (li.flags & 2u) {
            uint32_t ea = __float_as_uint(my_sbs[0]) >> 23;
            for (uint32_t p = 0; p < li.npass; ++p) {
                const uint32_t eb = __float_as_uint(my_sbs[p + 1u]) >> 23;
                if (ea == eb && ea >= 24u && ea < 255u && li.pass0 + p != 0u) {
                    mask |= 1u << p;
                    pred_e[p] = ea;
                }
                ea = eb;
            }
        }
        s_ahead = mask;
    }
    lds_barrier();
    const uint32_t ahead = s_ahead; // bit p: pass p gets its pair formed ahead, on the grid of binade pred_e[p]
    if (ahead) {
        uint32_t p = (uint32_t)__ffs((int)ahead) - 1u;
        fetch(q_first + p * SUPER);
        for (uint32_t left = ahead; left;) {
            left &= left - 1u;
            turn(v);
            const uint32_t pn = left ? (uint32_t)__ffs((int)left) - 1u : NONE;
            if (pn != NONE)
                fetch(q_first + pn * SUPER); // the next pass's entries travel while this one's pair is formed
            held = pn;
            const uint32_t e = pred_e[p];
            uint32_t a0, a1;
            bool tie;
            my_pair(__uint_as_float((277u - e) << 23), a0, a1, tie);
            uint32_t *w_a0 = w_a0s[flip], *w_a1 = w_a1s[flip];
            flip ^= 1u;
            uint32_t wt0, wt1;
            if (__ballot(tie) != 0ull) { // the runs of a wavefront composed in order: the last lane of an inclusive scan
                uint32_t i0 = a0, i1 = a1;
#pragma unroll
                for (uint32_t o = 1; o < (uint32_t)kWave; o <<= 1) {
                    const uint32_t f0 = (uint32_t)__shfl_up((int)i0, o, kWave), f1 = (uint32_t)__shfl_up((int)i1, o, kWave);
                    if (lane >= o)
                        compose(f0, f1, i0, i1, i0, i1);
                }
                wt0 = (uint32_t)__shfl((int)i0, kWave - 1, kWave), wt1 = (uint32_t)__shfl((int)i1, kWave - 1, kWave);
            } else {
                uint32_t x = a0;
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) {
                    x += (uint32_t)__shfl_xor((int)x, o, kWave);
                    x = x < SAT ? x : SAT;
                }
                wt0 = wt1 = x;
            }
            if (lane == 0)
                w_a0[wave] = wt0, w_a1[wave] = wt1;
            lds_barrier();
            if (tid == 0) {
                uint32_t t0 = 0, t1 = 0;
                for (uint32_t w = 0; w < NWV; ++w)
                    compose(t0, t1, w_a0[w], w_a1[w], t0, t1);
                sum_t0[p] = t0, sum_t1[p] = t1, sum_e[p] = e;
            }
            p = pn;
        }
    }
    // 2. the exact sum in front of my first term
    if (li.prev != NONE) {
        if (tid == 0) {
            unsigned long long w;
            while (((w = __hip_atomic_load(handoff + li.prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch)
                __builtin_amdgcn_s_sleep(8);
            s_bcast = __uint_as_float((uint32_t)w);
        }
        lds_barrier();
        S = s_bcast;
    }
    lds_barrier(); // (the pairs of step 1 are in LDS; s_bcast has been read by everyone before a pass writes it again)
    // 3. pass by pass
    for (uint32_t p = 0; p < li.npass; ++p) {
        if (tid == 0)
            my_sbs[p] = S; // what the next sweep predicts from
        const uint32_t sbits = __float_as_uint(S), e = sbits >> 23;
        if (sum_e[p] == e && e != 0u) { // the pair was formed on the grid S lies on
            const uint32_t J0 = (sbits & 0x7FFFFFu) | 0x800000u, cnt = (J0 & 1u) ? sum_t1[p] : sum_t0[p];
            if (J0 + cnt < (1u << 24)) {
                S = (float)(J0 + cnt) * __uint_as_float((e - 23u) << 23); // (J0 + count) ulp, exactly
                continue;
            }
        }
        if (held != p)
            fetch(q_first + p * SUPER);
        turn(v);
        held = NONE;
        if (p + 1u < li.npass && sum_e[p + 1u] == 0u) { // the next pass is walked as well: its entries travel meanwhile
            fetch(q_first + (p + 1u) * SUPER);           // (every barrier below waits for LDS traffic only)
            held = p + 1u;
        }
        seq_pass(li.pass0 + p == 0u);
    }
    // 4. hand the sum on, or finish the row
    if (tid == 0) {
        my_sbs[li.npass] = S;
        if (li.flags & 1u)
            st_agent(&group_err[item.group], pr_finalize(hub_rows[item.row0], S, base, damping, outdeg, scores, x_out));
        else
            __hip_atomic_store(handoff + me, ((unsigned long long)epoch << 32) | __float_as_uint(S), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    pb_err_fold(fold);
}

__global__ __launch_bounds__(1024) void pb_err_kernel(const double *__restrict__ bin_err, uint32_t B, double *__restrict__ err_out)
{
    __shared__ double red[1024 / kWave];
    double acc = 0.0;
    for (uint32_t b = threadIdx.x; b < B; b += 1024)
        acc += bin_err[b];
    const double total = block_sum<double, 1024 / kWave>(acc, red);
    if (threadIdx.x == 0)
        *err_out = total;
}

// Knobs that only measurements ever set (their A/B records: CHANGELOG.md) are compiled out of the product library: they read
// as their defaults there (round 6: the product library had grown ~80 environment variables).
#ifdef GM_MEASURE
#define pb_env_m(name, dflt) pb_env(name, dflt)
#define pb_getenv_m(name) getenv(name)
#else
#define pb_env_m(name, dflt) (dflt)
#define pb_getenv_m(name) (static_cast<const char *>(nullptr))
#endif
// tuning knobs (environment, read once): GM_PB_ABLATE measurement variants, GM_PB_CHUNK=<entries> phase-1
// workgroup size (0 = automatic), GM_PB_ORDER=0/1 longest-bin-first dispatch, GM_PB_RB=<log2 rows per bin>
int pb_env(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

unsigned pb_grid(uint64_t count)
{
    unsigned g = div_up(count, 256);
    return g > 256 * 32 ? 256 * 32 : (g ? g : 1);
}

int bits_for(uint64_t x) // smallest b with x <= 2^b
{
    int b = 0;
    while ((1ull << b) < x)
        ++b;
    return b;
}

template <class T> int scan_exclusive(const T *in, T *out, uint64_t count)
{
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, T(0), count, rocprim::plus<T>(), (hipStream_t)0));
    DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, T(0), count, rocprim::plus<T>(), (hipStream_t)0));
    return GM_OK;
}

int scan_inclusive_u32(const uint32_t *in, uint32_t *out, uint64_t count)
{
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, count, rocprim::plus<uint32_t>(), (hipStream_t)0));
    DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::inclusive_scan(tmp.p, tmp_bytes, in, out, count, rocprim::plus<uint32_t>(), (hipStream_t)0));
    return GM_OK;
}

int sort_keys_u64(DevBuf &keys, DevBuf &alt, uint64_t count, int begin_bit, int end_bit)
{
    rocprim::double_buffer<uint64_t> db(keys.as<uint64_t>(), alt.as<uint64_t>());
    size_t tmp_bytes = 0;
    GM_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, db, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    DevBuf tmp;
    GM_TRY(tmp.alloc(tmp_bytes));
    GM_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, db, count, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)0));
    GM_HIP(hipDeviceSynchronize());
    if (db.current() != keys.as<uint64_t>())
        std::swap(keys, alt);
    return GM_OK;
}

// *out = the first index whose bin field is >= bin (one thread: a binary search)
__global__ void pb_first_of_bin_kernel(const uint64_t *__restrict__ keys, uint32_t count, int shift, uint32_t mask, uint32_t bin,
                                       uint32_t *__restrict__ out)
{
    *out = (uint32_t)lower_bound_fn(0, count, (uint64_t)bin, [&](uint64_t i) { return (uint64_t)((uint32_t)(keys[i] >> shift) & mask); });
}

__global__ void pb_clear_bit_kernel(uint64_t *__restrict__ keys, uint32_t count, uint64_t bit)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
        keys[i] &= ~bit;
}

// Accumulate work items: one per bin, except that a bin more than twice the average size (a range of
// rows that attracts a large share of the edges, e.g. degree-sorted ids) is cut into slices so that no
// workgroup streams more than ~2x the average; longest first.  Built on the host from the B+1 bin
// boundaries (a few KiB).
// the rows pb_hublong_kernel sums (rows[0 .. count)) as items of GM_PB_LONG_PASSES passes (default 8: 65536 terms; 0 / >= 16: 16),
// row by row
int pb_make_long_items(PbPlan *pl, const std::vector<PbHubItem> &rows, uint32_t count)
{
    constexpr uint32_t SUPER = PB_LONG_WG * PB_LONG_PER;
    uint32_t per = (uint32_t)pb_env("GM_PB_LONG_PASSES", 8);
    per = per >= 1 && per <= PB_LONG_PMAX ? per : PB_LONG_PMAX;
    std::vector<PbLongItem> li;
    uint32_t sb = 0;
    for (uint32_t r = 0; r < count; ++r) {
        const uint32_t len = rows[r].q1 - rows[r].q0, passes = len ? (len + SUPER - 1u) / SUPER : 1u;
        for (uint32_t p0 = 0; p0 < passes; p0 += per) {
            const uint32_t np = passes - p0 < per ? passes - p0 : per;
            li.push_back(PbLongItem{r, p0, np, p0 ? (uint32_t)li.size() - 1u : 0xFFFFFFFFu, sb,
                                    (p0 + np == passes ? 1u : 0u) | (passes > per ? 2u : 0u)});
        }
        sb += passes + 1u;
    }
    pl->long_items_host = li;
    pl->n_long_items = (uint32_t)li.size();
    pl->long_sbs_len = sb;
    GM_TRY(pl->long_items.alloc((li.size() ? li.size() : 1) * sizeof(PbLongItem)));
    if (!li.empty())
        GM_HIP(hipMemcpy(pl->long_items.p, li.data(), li.size() * sizeof(PbLongItem), hipMemcpyHostToDevice));
    return GM_OK;
}

int pb_make_items(PbPlan *pl)
{
    const uint32_t Bv = pl->B + pl->G; // virtual bins of the streams
    std::vector<uint32_t> bv((size_t)Bv + 1), hv((size_t)Bv + 1, 0u);
    GM_HIP(hipMemcpy(bv.data(), pl->bin_v.p, bv.size() * 4, hipMemcpyDeviceToHost));
    if (pl->H) { // hv[b] = where bin b's hot edges (all its tiers) begin
        std::vector<uint32_t> cells((size_t)Bv * pl->T + 1);
        GM_HIP(hipMemcpy(cells.data(), pl->hbin_v.p, cells.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b <= Bv; ++b)
            hv[b] = cells[(size_t)b * pl->T];
    }
    const uint64_t total = (uint64_t)bv[pl->B] + hv[pl->B];
    uint64_t limit = 2 * (total / pl->B + 1);
    if (limit < 65536)
        limit = 65536;
    if (pb_env("GM_PB_SPLIT", 0) > 0)
        limit = (uint64_t)pb_env("GM_PB_SPLIT", 0);
    limit = (limit + 3) & ~3ull;
    std::vector<PbItem> items;
    uint32_t slots = 0;
    for (uint32_t b = 0; b < pl->B; ++b) {
        const uint32_t q0 = bv[b], q1 = bv[b + 1], g0 = hv[b], g1 = hv[b + 1];
        const uint64_t len = (uint64_t)(q1 - q0) + (g1 - g0);
        const uint32_t parts = len > limit ? (uint32_t)((len + limit - 1) / limit) : 1u;
        if (parts == 1) {
            items.push_back(PbItem{b, q0, q1, g0, g1, 1u, 0u, 0u});
            continue;
        }
        const uint32_t per = (((q1 - q0) + parts - 1) / parts + 3u) & ~3u;
        const uint32_t hper = (((g1 - g0) + parts - 1) / parts + 3u) & ~3u;
        for (uint32_t k = 0; k < parts; ++k) {
            uint64_t s = (uint64_t)q0 + (uint64_t)k * per, e = s + per;
            uint64_t hs = (uint64_t)g0 + (uint64_t)k * hper, he = hs + hper;
            s = s < q1 ? s : q1;
            e = e < q1 ? e : q1;
            hs = hs < g1 ? hs : g1;
            he = he < g1 ? he : g1;
            items.push_back(PbItem{b, (uint32_t)s, (uint32_t)e, (uint32_t)hs, (uint32_t)he, parts, slots, k});
        }
        slots += parts;
    }
    if (pb_env_m("GM_PB_ORDER", 1))
        std::stable_sort(items.begin(), items.end(),
                         [](const PbItem &a, const PbItem &c) {
                             return (uint64_t)(a.q1 - a.q0) + (a.h1 - a.h0) > (uint64_t)(c.q1 - c.q0) + (c.h1 - c.h0);
                         });
    pl->NI = (uint32_t)items.size();
    GM_TRY(pl->items.alloc(items.size() * sizeof(PbItem)));
    GM_HIP(hipMemcpy(pl->items.p, items.data(), items.size() * sizeof(PbItem), hipMemcpyHostToDevice));
    pl->items_host = items;
    pl->slots = slots;
    // hub groups, longest first: the long rows (one row per group, pb_hublong_kernel), then the groups pb_hubseq_kernel walks
    struct Tagged {
        PbHubItem item;
        bool is_long;
    };
    std::vector<Tagged> tagged;
    for (uint32_t g = 0; g < pl->G; ++g)
        tagged.push_back(Tagged{PbHubItem{bv[pl->B + g], bv[pl->B + g + 1], pl->hub_first_host[g + 1] - pl->hub_first_host[g],
                                          pl->hub_first_host[g], g},
                                g < pl->hub_long_host.size() && pl->hub_long_host[g] != 0});
    std::stable_sort(tagged.begin(), tagged.end(), [](const Tagged &a, const Tagged &c) { return a.item.q1 - a.item.q0 > c.item.q1 - c.item.q0; });
    std::stable_partition(tagged.begin(), tagged.end(), [](const Tagged &a) { return a.is_long; }); // each part longest first
    std::vector<PbHubItem> hubs;
    pl->G_long = 0;
    for (const Tagged &t : tagged) {
        hubs.push_back(t.item);
        pl->G_long += t.is_long ? 1u : 0u;
    }
    GM_TRY(pl->hub_items.alloc((hubs.size() ? hubs.size() : 1) * sizeof(PbHubItem)));
    if (!hubs.empty())
        GM_HIP(hipMemcpy(pl->hub_items.p, hubs.data(), hubs.size() * sizeof(PbHubItem), hipMemcpyHostToDevice));
    pl->hub_items_host = hubs;
    // error sums behind the B bins': one per hub group (written by pb_hubseq_kernel), then one per hub row (pb_hublong_kernel);
    // a slot nobody writes stays zero (pb_scratch_create)
    pl->err_slots = pl->G + pl->n_hub;
    // the rows pb_hublong_kernel sums: every row of every "long" group (one long row, or a thin group's few rows) as
    // {the group's stretch of the stream, the row's SLOT inside the group, its index in hub_rows, its error slot}, longest first
    std::vector<PbHubItem> rows;
    for (uint32_t g = 0; g < pl->G_long; ++g)
        for (uint32_t j = 0; j < hubs[g].nh; ++j)
            rows.push_back(PbHubItem{hubs[g].q0, hubs[g].q1, j, hubs[g].row0 + j, pl->G + hubs[g].row0 + j});
    pl->long_rows_host = rows;
    pl->n_long_rows = (uint32_t)rows.size();
    GM_TRY(pl->long_rows.alloc((rows.size() ? rows.size() : 1) * sizeof(PbHubItem)));
    if (!rows.empty())
        GM_HIP(hipMemcpy(pl->long_rows.p, rows.data(), rows.size() * sizeof(PbHubItem), hipMemcpyHostToDevice));
    return pb_make_long_items(pl, rows, (uint32_t)rows.size());
}

int pb_build(const gm_csr *csr, uint64_t x_len, PbPlan *pl)
{
    struct Site {
        int prev;
        Site() : prev(arena_site()) { arena_site() = 2; }
        ~Site() { arena_site() = prev; }
    } site_guard;
    const uint32_t n = (uint32_t)csr->n, m_all = (uint32_t)csr->m;
    uint32_t m = m_all; // becomes the number of cold (value-stream) edges once the hot ones are split off
    pl->n_local = n;
    pl->m = m_all;
    pl->x_len = x_len;
    pl->device = csr->device;
    // 16384-source tiles up to 2^26 sources; 32768 beyond, where the (tile, bin) segments of the smaller
    // tile drop below ~128 bytes (measured: scale 26 2.63 vs 2.71 ms for 14 vs 15, scale 27 5.90 vs 5.73 ms)
    pl->s_log = pb_env("GM_PB_SLOG", x_len > (1ull << 26) ? 15 : 14) >= PB_S_LOG_MAX ? PB_S_LOG_MAX : 14;
    // rows per bin: keep >= ~2048 bins so the accumulate kernel fills the chip, cap the LDS slice at 128 KiB
    int rb = bits_for(n) - 11;
    if (rb < 8)
        rb = 8;
    // ... but keep (tile, bin) segments >= ~64 entries (256-byte write runs): a tile sends about
    // m * S / x_len edges to this rank's rows, so at most that / 64 bins (matters for row slices of a
    // partitioned graph, where every source tile holds only 1/P of its edges)
    {
        const uint64_t per_tile = x_len ? ((uint64_t)m_all << pl->s_log) / x_len : 0;
        const uint64_t max_bins = per_tile / 64 > 1 ? per_tile / 64 : 1;
        while (rb < 14 && (((uint64_t)n + (1ull << rb) - 1) >> rb) > max_bins)
            ++rb;
    }
    if (rb > 14)
        rb = 14;
    if (pb_env("GM_PB_RB", 0) >= 8 && pb_env("GM_PB_RB", 0) <= 14)
        rb = pb_env("GM_PB_RB", 0);
    pl->rb = rb;
    pl->R = 1u << rb;
    pl->B = (uint32_t)(((uint64_t)n + pl->R - 1) >> rb);
    if (pl->B == 0)
        pl->B = 1;
    pl->NT = (uint32_t)((x_len + (1ull << pl->s_log) - 1) >> pl->s_log);
    if (pl->NT == 0)
        pl->NT = 1;
    const int sb = bits_for(x_len) < 1 ? 1 : bits_for(x_len);

    gm::PhaseTimer timer((hipStream_t)0); // GM_LOG=1: where the plan construction time goes
    // accumulator slots only for rows that have in-edges (RMAT: about half of the rows have none); hub rows
    // (summed in the reference's order, see the header) leave the ordinary bins and form hub groups
    pl->hub_deg = (uint32_t)pb_env("GM_PB_HUB_DEG", 4096);
    pl->hub_long = pb_env("GM_PB_HUB_LONG", 0) > 0 ? (uint32_t)pb_env("GM_PB_HUB_LONG", 0) : 8192u;
    GM_TRY(pl->cidx.alloc((size_t)(n ? n : 1) * 2));
    pl->Racc = 1;
    DevBuf pos_h; // hub rows before each row (kept until the keys are built)
    GM_TRY(pos_h.alloc_scratch(((size_t)n + 1) * 4));
    pl->hub_first_host.assign(1, 0u);
    if (n) {
        DevBuf flag, pos, bin_rows;
        GM_TRY(flag.alloc_scratch(((size_t)n + 1) * 4));
        GM_TRY(pos.alloc_scratch(((size_t)n + 1) * 4));
        GM_TRY(bin_rows.alloc((size_t)pl->B * 4));
        hipLaunchKernelGGL(pb_hubflag_kernel, dim3(pb_grid((uint64_t)n + 1)), dim3(256), 0, 0, csr->offsets, n, pl->hub_deg,
                           flag.as<uint32_t>());
        // rows below the threshold with many sources that have no in-edges themselves (GM_PB_HUB_LEAVES=<how many>, default 512; 0 = off):
        // see pb_leafflag_kernel.  A whole graph reads a source's in-degree off its own offsets; a partition slice needs the flags
        // of gm_csr_set_source_flags (the C++ partitioned front hands them over, multi.hip) — without them the rule is skipped
        // there, and where it would have flagged a row the slice's bits are not the single engine's.
        const uint32_t leaf_t = hub_leaves_threshold();
        const bool have_flags = csr->source_flags_len >= x_len && csr->source_flags.p;
        if ((x_len == n || have_flags) && leaf_t && pl->hub_deg > leaf_t)
            hipLaunchKernelGGL(pb_leafflag_kernel, dim3(pb_grid(n)), dim3(256), 0, 0, csr->offsets, csr->targets, n, pl->hub_deg, leaf_t,
                               have_flags ? csr->source_flags.as<uint8_t>() : (const uint8_t *)nullptr, csr->source_flags_len,
                               flag.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_TRY(scan_exclusive<uint32_t>(flag.as<uint32_t>(), pos_h.as<uint32_t>(), (uint64_t)n + 1));
        GM_HIP(hipMemcpy(&pl->n_hub, pos_h.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost));
        timer.done("pb plan: - hub flags + scan");
        if (pl->n_hub) {
            // hub groups: consecutive hub rows, cut at PB_HUB_MAX rows or about one ordinary bin's worth of
            // terms (so that a group's (tile, group) segments are as long as a bin's)
            DevBuf hub_degs;
            GM_TRY(pl->hub_rows.alloc((size_t)pl->n_hub * 4));
            GM_TRY(hub_degs.alloc((size_t)pl->n_hub * 4));
            hipLaunchKernelGGL(pb_hub_rows_kernel, dim3(pb_grid(n)), dim3(256), 0, 0, csr->offsets, pos_h.as<uint32_t>(), flag.as<uint32_t>(),
                               n, pl->hub_rows.as<uint32_t>(), hub_degs.as<uint32_t>());
            GM_HIP(hipGetLastError());
            std::vector<uint32_t> degs(pl->n_hub);
            GM_HIP(hipMemcpy(degs.data(), hub_degs.p, (size_t)pl->n_hub * 4, hipMemcpyDeviceToHost));
            pl->hub_degs_host = degs;
            // Are the hub rows' lists ascending?  Then the order the value stream delivers their terms in IS the CSR order
            // (Sorted / Deduplicated layouts).  If not (CsrLayout::Unsorted), the rows are summed through a per-term index in
            // CSR order (pb_hublong_kernel<true>; GM_PB_HUB_CSR=0: in source order all the same, round 4's behaviour).
            {
                DevBuf unsorted;
                GM_TRY(unsorted.alloc(4));
                GM_HIP(hipMemset(unsorted.p, 0, 4));
                hipLaunchKernelGGL(pb_hub_sorted_kernel, dim3(pb_grid((uint64_t)pl->n_hub * kWave)), dim3(256), 0, 0, csr->offsets,
                                   csr->targets, pl->hub_rows.as<uint32_t>(), pl->n_hub, unsorted.as<uint32_t>());
                GM_HIP(hipGetLastError());
                uint32_t flag_host = 0;
                GM_HIP(hipMemcpy(&flag_host, unsorted.p, 4, hipMemcpyDeviceToHost));
                pl->hub_csr = flag_host && pb_env("GM_PB_HUB_CSR", 1) ? 1u : 0u;
            }
            uint64_t target = (uint64_t)m_all / pl->B;
            if (target < 65536)
                target = 65536;
            if (pb_env_m("GM_PB_HUB_GROUP", 0) > 0)
                target = (uint64_t)pb_env_m("GM_PB_HUB_GROUP", 0);
            // Which rows are "long"?  A row walked by one lane of pb_hubseq_kernel costs ~6 ns per term, all of it latency (a
            // chain of dependent adds); pb_hublong_kernel sums a row of 10^4 terms in ~15 us and one of 10^6 in ~100, but
            // keeps a 512-thread workgroup busy with vector work.  So: rows of at least 8192 in-edges, but not more than
            // about two per CU — the threshold moves up until at most 512 rows are long.  GM_PB_HUB_LONG fixes it.
            if (pb_env("GM_PB_HUB_LONG", 0) <= 0) {
                std::vector<uint32_t> sorted(degs);
                std::sort(sorted.begin(), sorted.end(), [](uint32_t a, uint32_t c) { return a > c; });
                pl->hub_long = 8192;
                if (sorted.size() > 512 && sorted[512] + 1u > pl->hub_long)
                    pl->hub_long = sorted[512] + 1u;
            }
            // The hub rows' index space, regrouped for the SLICES of a partitioned graph (round 6; GM_PB_HUB_REGROUP=0 / 1: never /
            // always): the rows one lane walks come first, in ascending order, the long ones — a group of their own each — behind
            // them.  Groups are RANGES of that index space, and in ascending row order a long row between two walked ones ended a
            // group: in a rank's slice, where 22 % of the hub rows are long, the walked groups held runs of 4.6 rows (an 8-way rank
            // at scale 26: 2,310 hub rows in 902 groups, one pb_hubseq_kernel workgroup each beside the accumulate kernel, 60 of
            // 64 lanes of every walk idle; regrouped: 630 groups).  A row's sum does not depend on its group: same bits (the whole
            // GPU suite with it on everywhere).  Measured (tools/runs/r06_call23.sh, alternating processes): emulated ranks 0 / 6
            // of 8 0.650 / 0.704 -> 0.595 / 0.609 ms, rank 0 of 4 1.056 -> 0.998; the whole graph at scale 26 2.74-2.78 -> 2.71-2.74
            // (1530 -> 1266 groups), but at scale 22 0.203-0.206 -> 0.214-0.217 (507 -> 378 groups: the fuller groups' walks
            // outlast the 116 us accumulate kernel they run beside) — hence slices only.
            if (pb_env("GM_PB_HUB_REGROUP", x_len != n ? 1 : 0)) {
                std::vector<uint32_t> rows_old(pl->n_hub), order;
                GM_HIP(hipMemcpy(rows_old.data(), pl->hub_rows.p, (size_t)pl->n_hub * 4, hipMemcpyDeviceToHost));
                order.reserve(pl->n_hub);
                for (uint32_t h = 0; h < pl->n_hub; ++h)
                    if (degs[h] < pl->hub_long)
                        order.push_back(h);
                for (uint32_t h = 0; h < pl->n_hub; ++h)
                    if (degs[h] >= pl->hub_long)
                        order.push_back(h);
                std::vector<uint32_t> rows_new(pl->n_hub), degs_new(pl->n_hub);
                for (uint32_t k = 0; k < pl->n_hub; ++k)
                    rows_new[k] = rows_old[order[k]], degs_new[k] = degs[order[k]];
                GM_HIP(hipMemcpy(pl->hub_rows.p, rows_new.data(), (size_t)pl->n_hub * 4, hipMemcpyHostToDevice));
                degs = degs_new;
                pl->hub_degs_host = degs;
                // pos_h[row] = the row's index in that space (what the key kernel looks a hub row up by)
                hipLaunchKernelGGL(pb_hub_repos_kernel, dim3(pb_grid(pl->n_hub)), dim3(256), 0, 0, pl->hub_rows.as<uint32_t>(), pl->n_hub,
                                   pos_h.as<uint32_t>());
                GM_HIP(hipGetLastError());
            }
            // ... and in a slice the walked groups are made SMALLER, so that there is about one per CU (GM_PB_HUB_SLICE_GROUPS, default
            // 224; 0: the whole-graph target): a group's walk is a chain whose length is its terms (one lane per row, a 2040-term
            // block at a time: ~1.6 us per block), and a slice's accumulate phase is short — at 8 ranks the full-size groups' walks
            // (207-217 us) outlasted BOTH accumulate pieces of the rank (106 + 75 us) and the first piece's exchange waited 125 us
            // for them (profiles/r06_rank_of_8_sweep_timeline.txt).
            if (x_len != n) {
                const uint64_t want_groups = (uint64_t)pb_env("GM_PB_HUB_SLICE_GROUPS", 224);
                uint64_t walked_terms = 0;
                for (uint32_t h = 0; h < pl->n_hub; ++h)
                    walked_terms += degs[h] < pl->hub_long ? degs[h] : 0u;
                if (want_groups) {
                    uint64_t t = walked_terms / want_groups;
                    t = t < 32768 ? 32768 : t;
                    target = t < target ? t : target;
                }
            }
            uint64_t acc_edges = 0;
            uint32_t count = 0;
            bool prev_long = false;
            for (uint32_t i = 0; i < pl->n_hub; ++i) {
                pl->hub_edges += degs[i];
                const bool is_long = degs[i] >= pl->hub_long; // a group of its own: pb_hublong_kernel
                pl->long_terms += is_long ? degs[i] : 0u;
                if (count && (count == PB_HUB_MAX || acc_edges + degs[i] > target || is_long || prev_long)) {
                    pl->hub_first_host.push_back(i);
                    pl->hub_long_host.push_back(prev_long ? 1 : 0);
                    acc_edges = 0, count = 0;
                }
                acc_edges += degs[i];
                ++count;
                prev_long = is_long;
            }
            pl->hub_long_host.push_back(prev_long ? 1 : 0);
            pl->hub_first_host.push_back(pl->n_hub);
            pl->G = (uint32_t)pl->hub_first_host.size() - 1;
            // THIN groups (GM_PB_HUB_THIN=<rows>, default 0 = off; round 5): a group of few rows leaves most lanes of
            // pb_hubseq_kernel's walking wavefront idle — in a rank's slice of a partitioned graph, where hub rows are seldom
            // neighbours, that is most groups (an 8-way rank at scale 26: 2,310 hub rows in 902 groups).  Such a group can be
            // summed by pb_hublong_kernel instead, one item list per ROW, each reading the group's stretch of the stream and
            // taking the entries of its slot.  Bit-identical (tests/test_gpu_hub_adversarial.py) and MEASURED SLOWER with 4
            // (tools/runs/r05_call20.sh, alternating processes): rank 0 / 1 of 8 0.586 / 0.529 against 0.565 / 0.500 ms, scale
            // 26 2.70 against 2.65, scale 22 0.210-0.226 against 0.207 — the 512-thread workgroups of twice as many rows cost the
            // accumulate kernel beside them more than the idle lanes of the walks did: off.
            {
                const uint32_t thin = (uint32_t)pb_env("GM_PB_HUB_THIN", 0);
                for (uint32_t g = 0; g < pl->G; ++g) {
                    const uint32_t nh = pl->hub_first_host[g + 1] - pl->hub_first_host[g];
                    if (!pl->hub_long_host[g] && nh <= thin) {
                        pl->hub_long_host[g] = 1;
                        for (uint32_t h = pl->hub_first_host[g]; h < pl->hub_first_host[g + 1]; ++h)
                            pl->long_terms += degs[h];
                    }
                }
            }
        }
        GM_TRY(pl->hub_first.alloc(pl->hub_first_host.size() * 4));
        GM_HIP(hipMemcpy(pl->hub_first.p, pl->hub_first_host.data(), pl->hub_first_host.size() * 4, hipMemcpyHostToDevice));
        timer.done("pb plan: - hub groups");
        hipLaunchKernelGGL(pb_rowflag_kernel, dim3(pb_grid((uint64_t)n + 1)), dim3(256), 0, 0, csr->offsets, n, flag.as<uint32_t>());
        GM_HIP(hipGetLastError());
        GM_TRY(scan_exclusive<uint32_t>(flag.as<uint32_t>(), pos.as<uint32_t>(), (uint64_t)n + 1));
        hipLaunchKernelGGL(pb_cidx_kernel, dim3(pb_grid(n)), dim3(256), 0, 0, csr->offsets, pos.as<uint32_t>(), flag.as<uint32_t>(), n, rb,
                           pl->cidx.as<uint16_t>(), bin_rows.as<uint32_t>());
        GM_HIP(hipGetLastError());
        std::vector<uint32_t> rows(pl->B);
        GM_HIP(hipMemcpy(rows.data(), bin_rows.p, (size_t)pl->B * 4, hipMemcpyDeviceToHost));
        timer.done("pb plan: - row flags + scan + slots");
        uint32_t mx = 1;
        for (uint32_t v : rows)
            mx = v > mx ? v : mx;
        pl->Racc = (mx + 63u) & ~63u;
        if (pl->Racc > pl->R)
            pl->Racc = pl->R;
        if (pb_env_m("GM_PB_COMPACT", 1) == 0)
            pl->Racc = pl->R; // keep the slot numbering, size the LDS as if every row had one
    } else {
        GM_TRY(pl->hub_first.alloc(4));
        GM_HIP(hipMemset(pl->hub_first.p, 0, 4));
    }

    const uint32_t Bv = pl->B + pl->G; // virtual bins of the streams: the ordinary bins, then the hub groups
    const int bin_bits = bits_for(Bv) < 1 ? 1 : bits_for(Bv);
    GM_CHECK(bin_bits + sb + rb + 1 <= 64, GM_ERR_RANGE, "pb_build: key does not fit 64 bits");

    // hot table size: what is left of the LDS beside the accumulators (2 workgroups per CU when the
    // accumulators are <= 64 KiB, else 1)
    uint32_t H = 0, H_single = 0; // sources per tier with two tables in LDS / with one
    {
        const size_t acc_bytes = (size_t)pl->Racc * 8 + 16; // + the accumulator the padding entries go to
        int wgs = acc_bytes > 65536 ? 1 : 2; // accumulate workgroups per CU the LDS request should allow
        if (pb_env_m("GM_PB_WGS", 0) == 1 || (pb_env_m("GM_PB_WGS", 0) == 0 && acc_bytes > 32768))
            wgs = 1;
        // static LDS of the accumulate kernel: PB_ACC_STATIC; with hub groups, room for one pb_hubseq_kernel workgroup
        // and one pb_hublong_kernel workgroup (21.5 KiB together) beside the accumulate workgroup(s) of a CU
        size_t hub_room = (pl->G && pb_env("GM_PB_HUB_FORK", 1)) ? PB_HUB_ROOM : 0;
        if (pb_env_m("GM_PB_HUB_ROOM", -1) >= 0) // measurement: LDS left free beside an accumulate workgroup
            hub_room = (size_t)pb_env_m("GM_PB_HUB_ROOM", -1);
        const size_t budget = wgs == 1 ? (163840 - hub_room - PB_ACC_STATIC - acc_bytes)
                                       : ((163840 - hub_room) / 2 - PB_ACC_STATIC - acc_bytes);
        H = (uint32_t)(budget / 4) & ~63u;
        if (H > 32768)
            H = 32768;
        const int cap = pb_env("GM_PB_HOT", -1);
        if (cap >= 0 && (uint32_t)cap < H)
            H = (uint32_t)cap & ~63u;
        H_single = H;
        if (pb_env("GM_PB_TIERS", 0) != 1) { // two tables in LDS: a tier is half of the room
            H = (H / 2) & ~63u;
            if (H > 16384)
                H = 16384;
        }
    }

    // tiers of hot sources (GM_PB_TIERS; measured at RMAT scale 26: see DESIGN 4.1)
    uint32_t T = (uint32_t)(pb_env("GM_PB_TIERS", 0) > 0 ? pb_env("GM_PB_TIERS", 0) : PB_TIERS_DEFAULT); // 0: automatic, up to the default
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    if (H == 0)
        T = 1;
    GM_TRY(pl->bin_v.alloc(((size_t)Bv + 1) * 4));
    GM_TRY(pl->hbin_v.alloc(((size_t)Bv * T + 1) * 4));
    GM_TRY(pl->tile_p.alloc(((size_t)pl->NT + 1) * 4));
    GM_HIP(hipMemset(pl->hbin_v.p, 0, ((size_t)Bv * T + 1) * 4));
    GM_TRY(pl->hot_ent.alloc(16));
    GM_TRY(pl->hot_ids.alloc((size_t)(H ? (size_t)H * T : 1) * 4));
    pl->T = T;
    if (m_all == 0) {
        GM_TRY(pl->p2_dst.alloc(16));
        GM_HIP(hipMemset(pl->bin_v.p, 0, ((size_t)Bv + 1) * 4));
        GM_HIP(hipMemset(pl->tile_p.p, 0, ((size_t)pl->NT + 1) * 4));
        pl->H = 0;
        GM_TRY(pb_make_items(pl));
        pl->NW = 0;
        return GM_OK;
    }

    timer.done("pb plan: accumulator slots");
    // ---- hot sources: the H most frequent source ids (>= 2 edges) of this rank's edges ----------------
    DevBuf hot_rank, hot_blk;
    int filter_bits = pb_env_m("GM_PB_FILTER_BITS", PB_FILTER_DEFAULT); // log2 of the block filter's size in bits
    filter_bits = filter_bits < 10 ? 10 : (filter_bits > PB_FILTER_BITS ? PB_FILTER_BITS : filter_bits);
    const int fshift = sb > filter_bits ? sb - filter_bits : 0;
    const uint32_t filter_words = (uint32_t)(((x_len >> fshift) + 32) / 32);
    if (H) {
        DevBuf cnt, ckeys, calt, n_keys;
        GM_TRY(cnt.alloc_big((size_t)x_len * 4));
        GM_TRY(ckeys.alloc_big((size_t)x_len * 8));
        GM_TRY(calt.alloc_big((size_t)x_len * 8));
        GM_TRY(n_keys.alloc(4));
        GM_TRY(hot_rank.alloc_big((size_t)x_len * 4));
        GM_TRY(hot_blk.alloc((size_t)filter_words * 4));
        GM_HIP(hipMemset(cnt.p, 0, (size_t)x_len * 4));
        GM_HIP(hipMemset(n_keys.p, 0, 4));
        GM_HIP(hipMemset(hot_blk.p, 0, hot_blk.bytes));
        GM_HIP(hipMemset(hot_rank.p, 0xFF, (size_t)x_len * 4));
        // one edge in 32 beyond 2^28 edges (one in 8 beyond 2^26): a source of the hot set has thousands of edges, and
        // the 134 M random atomics of a 1/8 sample were 6 ms of the plan at scale 26
        const uint32_t sample_step = m_all > (1u << 28) ? 32u : m_all > (1u << 26) ? 8u : 1u;
        hipLaunchKernelGGL(pb_count_sources_kernel, dim3(pb_grid(m_all / sample_step + 1)), dim3(256), 0, 0, csr->targets,
                           m_all, sample_step, cnt.as<uint32_t>());
        hipLaunchKernelGGL(pb_count_keys_kernel, dim3(pb_grid(x_len)), dim3(256), 0, 0, cnt.as<uint32_t>(), (uint32_t)x_len,
                           ckeys.as<uint64_t>(), n_keys.as<uint32_t>());
        GM_HIP(hipGetLastError());
        uint32_t candidates = 0;
        GM_HIP(hipMemcpy(&candidates, n_keys.p, 4, hipMemcpyDeviceToHost));
        if (candidates)
            GM_TRY(sort_keys_u64(ckeys, calt, candidates, 0, 64)); // a few million keys of the 67 M sources at scale 26
        H = H < candidates ? H : candidates;
        if (H) {
            const uint64_t cap = (uint64_t)H * T; // whole tiers of H sources, the last one possibly short
            pl->Htot = (uint32_t)(cap < candidates ? cap : candidates);
            T = (pl->Htot + H - 1) / H;
            if (T > 1 && pb_env("GM_PB_TIERS", 0) <= 0) {
                // How many tiers pay?  A tier costs every accumulate workgroup a barrier and a mostly idle pass over its
                // few edges (~2 us), and saves 8 streamed bytes per edge it takes over: measured break-even ~5000 edges
                // per (bin, tier) cell (scale 22: one tier; scale 24: sixteen; scale 26: eight).  The candidates' keys
                // hold their sampled edge counts.
                std::vector<uint64_t> top(pl->Htot);
                GM_HIP(hipMemcpy(top.data(), ckeys.p, (size_t)pl->Htot * 8, hipMemcpyDeviceToHost));
                const double need = 6000.0 * pl->B;
                uint32_t keep = 1;
                for (uint32_t t = 1; t < T; ++t) {
                    double edges = 0;
                    for (uint32_t k = t * H; k < pl->Htot && k < (t + 1) * H; ++k)
                        edges += (double)(uint32_t)~(uint32_t)(top[k] >> 32) * sample_step;
                    if (edges < need)
                        break;
                    keep = t + 1;
                }
                T = keep;
                if (T == 1) // one table can use the whole room
                    H = H_single < candidates ? H_single : candidates;
                pl->Htot = (uint32_t)(((uint64_t)H * T) < candidates ? (uint64_t)H * T : candidates);
            }
            if (T == 1 && pb_env_m("GM_PB_HOT_TRIM", 1) && H > 1024) {
                // ONE table: every accumulate workgroup stages all of it before its first hot edge (4 H bytes from L2 and a
                // barrier), so a source belongs in it only if its edges pay for that.  The same break-even as a tier's
                // (6000 edges per bin for ~10,000 sources): a source with fewer edges than half the number of bins is left
                // to the value stream.  Measured at RMAT scale 22 (tools/runs/r05_call14.sh, 2048 bins): the whole room
                // (15,360 sources, the last ones ~800 edges each) 0.2155 ms per sweep, 8,192 sources (~1,100) 0.2069, 4,096
                // (~1,900) 0.2096.
                std::vector<uint64_t> top(H);
                GM_HIP(hipMemcpy(top.data(), ckeys.p, (size_t)H * 8, hipMemcpyDeviceToHost));
                const double least = 0.5 * pl->B;
                uint32_t keep = H;
                while (keep > 1024 && (double)(uint32_t)~(uint32_t)(top[keep - 1] >> 32) * sample_step < least)
                    keep -= 64;
                H = keep & ~63u;
                pl->Htot = H;
            }
            pl->T = T;
            hipLaunchKernelGGL(pb_hot_select_kernel, dim3(div_up(pl->Htot, 256)), dim3(256), 0, 0, ckeys.as<uint64_t>(), pl->Htot,
                               pl->hot_ids.as<uint32_t>(), hot_rank.as<uint32_t>(), hot_blk.as<uint32_t>(), fshift);
        }
        GM_HIP(hipGetLastError());
        GM_HIP(hipDeviceSynchronize());
        if (H == 0) {
            hot_rank.release();
            hot_blk.release();
        }
    }
    pl->H = H;
    if (H == 0)
        pl->T = T = 1, pl->Htot = 0;
    timer.done("pb plan: hot source selection");

    DevBuf keys, kalt;
    GM_TRY(keys.alloc_big((size_t)m_all * 8)); // arena.hip: large buffers never go back to the driver while the
    GM_TRY(kalt.alloc_big((size_t)m_all * 8)); // process may need them again (no hipMalloc stall after large frees)
    timer.done("pb plan: - key buffers (2 x %.1f GB)", (double)m_all * 8 / 1e9);
    const int hot_bit = bin_bits + sb; // the flag bit of a hot edge: the highest sorted bit
    // hub groups walked by pb_hubseq_kernel take their terms from hot sources off the value stream (GM_PB_HUB_HOT=0: not): the
    // flag of such an edge sits above the slot
    const bool hub_hot = H && pl->G > 0 && pb_env("GM_PB_HUB_HOT", 1) != 0 && pl->Htot < (1u << 18) && !pl->hub_csr; // (CSR order: every term in the stream)
    const uint64_t hh_bit = hub_hot ? 1ull : 0ull; // (a switch for pb_keys_kernel: the key itself carries no extra bit)
    DevBuf group_long;
    {
        std::vector<uint32_t> gl(pl->G ? pl->G : 1, 0u);
        for (uint32_t g = 0; g < pl->G && g < pl->hub_long_host.size(); ++g)
            gl[g] = pl->hub_long_host[g] ? 1u : 0u;
        GM_TRY(group_long.alloc(gl.size() * 4));
        GM_HIP(hipMemcpy(group_long.p, gl.data(), gl.size() * 4, hipMemcpyHostToDevice));
    }
    {
        const size_t lds = H ? (size_t)filter_words * 4 : 0;
        GM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_keys_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((4u << PB_FILTER_BITS) / 32 + 64)));
        unsigned kg = div_up(n, PB_KEYS_BLOCK);
        kg = kg > 512 ? 512 : (kg ? kg : 1); // persistent workgroups: each stages the filter once
        hipLaunchKernelGGL(pb_keys_kernel, dim3(kg), dim3(PB_KEYS_BLOCK), lds, 0, csr->offsets, csr->targets, n, rb, sb, bin_bits,
                           H ? hot_blk.as<uint32_t>() : (const uint32_t *)nullptr, H ? filter_words : 0u, fshift,
                           hot_rank.as<uint32_t>(), pl->cidx.as<uint16_t>(), pos_h.as<uint32_t>(), pl->hub_first.as<uint32_t>(),
                           pl->B, pl->G, group_long.as<uint32_t>(), hh_bit, keys.as<uint64_t>());
    }
    GM_HIP(hipGetLastError());
    timer.done("pb plan: - edge keys");
    // the slot sits above the sorted bits (rocPRIM's radix sort was measured 14x slower with a non-zero BEGIN bit at
    // this size, so the unsorted field is at the top, not at the bottom)
    GM_TRY(sort_keys_u64(keys, kalt, m_all, 0, H ? hot_bit + 1 : hot_bit));
    kalt.release();
    if (!hub_hot)
        hot_rank.release(); // (with hub_hot: kept for the hot records of the hub groups, pb_hubseq_layout_kernel)
    hot_blk.release();
    pos_h.release();
    timer.done("pb plan: edge keys + sort");

    // hot keys (top bit set) sit behind the cold ones: the ordinary ones (bins < B) ordered by (bin, rank = tier-major, row),
    // then the hub groups' (bins >= B) ordered by (group, source)
    const uint64_t *ckeys = keys.as<uint64_t>(); // the cold keys: the value stream's entries
    const uint64_t *hk = nullptr;                // the hub groups' hot keys
    uint32_t mhh = 0;
    if (H) {
        DevBuf split;
        GM_TRY(split.alloc(4 * 4));
        hipLaunchKernelGGL(pb_bounds_kernel, dim3(pb_grid(m_all)), dim3(256), 0, 0, keys.as<uint64_t>(), m_all, hot_bit, 2u,
                           split.as<uint32_t>(), 1u);
        GM_HIP(hipGetLastError());
        GM_HIP(hipMemcpy(&m, split.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost));
        uint32_t mh = m_all - m;
        if (mh && hub_hot) { // where the hub groups' hot keys begin
            uint32_t first_hub = mh;
            hipLaunchKernelGGL(pb_first_of_bin_kernel, dim3(1), dim3(1), 0, 0, keys.as<uint64_t>() + m, mh, sb,
                               (uint32_t)((1ull << bin_bits) - 1ull), pl->B, split.as<uint32_t>() + 3);
            GM_HIP(hipGetLastError());
            GM_HIP(hipMemcpy(&first_hub, split.as<uint32_t>() + 3, 4, hipMemcpyDeviceToHost));
            mhh = mh - first_hub;
            mh = first_hub;
            hk = keys.as<uint64_t>() + m + mh;
            if (mhh && m == 0) {
                // nothing is left of the value stream (a small graph whose every source is hot): the hub groups' hot keys
                // go back to being cold ones — as a block they are sorted by (virtual bin, source) like cold keys
                hipLaunchKernelGGL(pb_clear_bit_kernel, dim3(pb_grid(mhh)), dim3(256), 0, 0, keys.as<uint64_t>() + mh, mhh,
                                   1ull << hot_bit);
                GM_HIP(hipGetLastError());
                ckeys = keys.as<uint64_t>() + mh;
                m = mhh, mhh = 0, hk = nullptr;
            }
        }
        if (mh) {
            const uint64_t *hkeys = keys.as<uint64_t>() + (ckeys == keys.as<uint64_t>() ? m : 0u);
            DevBuf hstart, hpad;
            const uint32_t cells = Bv * T; // (bin, tier) cells
            GM_TRY(hstart.alloc(((size_t)cells + 1) * 4));
            GM_TRY(hpad.alloc(((size_t)cells + 1) * 4));
            hipLaunchKernelGGL(pb_hot_bounds_kernel, dim3(pb_grid(mh)), dim3(256), 0, 0, hkeys, mh, sb, bin_bits, H, T, cells,
                               hstart.as<uint32_t>());
            // 2-byte records (GM_PB_HOT16=0: 4-byte ones): the row slot must leave a 14-bit code free for the fillers.
            // Measured at scale 26 (tools/runs/r04_call57.sh / r04_call58.sh): 0.54 GB fewer bytes per sweep and 0.5 GB less
            // plan, the accumulate kernel 1552 against 1572 us — its hot phase is not what the memory system bounds.
            pl->hot16 = pb_env("GM_PB_HOT16", 1) && pl->Racc < 16384u && H <= 16384u ? 1u : 0u;
            if (pl->hot16) {
                DevBuf entries, epos;
                GM_TRY(entries.alloc_scratch(((size_t)mh + 1) * 4));
                GM_TRY(epos.alloc_scratch(((size_t)mh + 1) * 4));
                hipLaunchKernelGGL(pb_h16_count_kernel, dim3(pb_grid((uint64_t)mh + 1)), dim3(256), 0, 0, hkeys, mh, sb, bin_bits, H,
                                   T, entries.as<uint32_t>());
                GM_HIP(hipGetLastError());
                GM_TRY(scan_exclusive<uint32_t>(entries.as<uint32_t>(), epos.as<uint32_t>(), (uint64_t)mh + 1));
                hipLaunchKernelGGL(pb_h16_sizes_kernel, dim3(pb_grid((uint64_t)cells + 1)), dim3(256), 0, 0, hstart.as<uint32_t>(),
                                   epos.as<uint32_t>(), cells, hpad.as<uint32_t>());
                GM_HIP(hipGetLastError());
                GM_TRY(scan_exclusive<uint32_t>(hpad.as<uint32_t>(), pl->hbin_v.as<uint32_t>(), (uint64_t)cells + 1));
                uint32_t Mh = 0;
                GM_HIP(hipMemcpy(&Mh, pl->hbin_v.as<uint32_t>() + cells, 4, hipMemcpyDeviceToHost));
                pl->Mh = Mh;
                GM_TRY(pl->hot_ent.alloc_big((size_t)Mh * 2 + 16, 0x407E, 4, 0, ~0ull, 0, 4));
                GM_TRY(pl->hot_base.alloc(((size_t)Mh / PB_H16_LANE + 2) * 2));
                GM_HIP(hipMemset(pl->hot_base.p, 0, pl->hot_base.bytes));
                hipLaunchKernelGGL(pb_h16_pattern_kernel, dim3(pb_grid((uint64_t)Mh + 8)), dim3(256), 0, 0, pl->hot_ent.as<uint16_t>(),
                                   (uint64_t)Mh + 8, (uint16_t)(pl->Racc << 2));
                hipLaunchKernelGGL(pb_h16_fill_kernel, dim3(pb_grid(mh)), dim3(256), 0, 0, hkeys, mh, hstart.as<uint32_t>(),
                                   epos.as<uint32_t>(), pl->hbin_v.as<uint32_t>(), sb, bin_bits, H, T, pl->Racc,
                                   pl->hot_ent.as<uint16_t>(), pl->hot_base.as<uint16_t>());
                GM_HIP(hipGetLastError());
                GM_HIP(hipDeviceSynchronize());
            } else {
            hipLaunchKernelGGL(pb_pad4_sizes_kernel, dim3(pb_grid((uint64_t)cells + 1)), dim3(256), 0, 0,
                               hstart.as<uint32_t>(), cells, hpad.as<uint32_t>());
            GM_HIP(hipGetLastError());
            GM_TRY(scan_exclusive<uint32_t>(hpad.as<uint32_t>(), pl->hbin_v.as<uint32_t>(), (uint64_t)cells + 1));
            uint32_t Mh = 0;
            GM_HIP(hipMemcpy(&Mh, pl->hbin_v.as<uint32_t>() + cells, 4, hipMemcpyDeviceToHost));
            pl->Mh = Mh;
            GM_TRY(pl->hot_ent.alloc_big((size_t)Mh * 4, 0x407E, 4, 0, ~0ull, 0, 4));
            GM_HIP(hipMemset(pl->hot_ent.p, 0xFF, (size_t)Mh * 4));
            hipLaunchKernelGGL(pb_hot_fill_kernel, dim3(pb_grid(mh)), dim3(256), 0, 0, hkeys, mh, hstart.as<uint32_t>(),
                               pl->hbin_v.as<uint32_t>(), sb, bin_bits, H, T, pl->hot_ent.as<uint32_t>());
            GM_HIP(hipGetLastError());
            GM_HIP(hipDeviceSynchronize());
            }
        }
    }
    pl->Mhh = mhh;
    if (m == 0) { // every edge is hot
        GM_TRY(pl->p2_dst.alloc(16));
        GM_HIP(hipMemset(pl->bin_v.p, 0, ((size_t)Bv + 1) * 4));
        GM_HIP(hipMemset(pl->tile_p.p, 0, ((size_t)pl->NT + 1) * 4));
        GM_TRY(pb_make_items(pl));
        pl->NW = 0;
        return GM_OK;
    }
    const unsigned gm_ = pb_grid(m);

    timer.done("pb plan: hot edge stream");
    // (bin, tile) segments of the sorted entries
    DevBuf flag, segid;
    GM_TRY(flag.alloc_big((size_t)m * 4));
    GM_TRY(segid.alloc_big((size_t)m * 4));
    hipLaunchKernelGGL(pb_flags_kernel, dim3(gm_), dim3(256), 0, 0, ckeys, m, bin_bits, sb, pl->s_log,
                       flag.as<uint32_t>());
    GM_HIP(hipGetLastError());
    GM_TRY(scan_inclusive_u32(flag.as<uint32_t>(), segid.as<uint32_t>(), m));
    uint32_t NS = 0;
    GM_HIP(hipMemcpy(&NS, segid.as<uint32_t>() + (m - 1), 4, hipMemcpyDeviceToHost));
    pl->NS = NS;
    GM_CHECK((uint64_t)m + 3ull * NS + (uint64_t)pl->NT * PB_WBLK < (1ull << 32), GM_ERR_RANGE,
             "pb_build: padded streams exceed 2^32 entries");

    DevBuf vstart, segkey, segkalt, segval, segbin;
    const int jb = bits_for(NS) < 1 ? 1 : bits_for(NS), tile_bits = bits_for(pl->NT) < 1 ? 1 : bits_for(pl->NT);
    const int tile_shift = bin_bits + jb;
    GM_CHECK(tile_bits + tile_shift <= 64, GM_ERR_RANGE, "pb_build: %u tiles x %u bins x %u segments do not fit a 64-bit key",
             pl->NT, Bv, NS);
    GM_TRY(vstart.alloc_scratch((size_t)NS * 4));
    GM_TRY(segkey.alloc_scratch((size_t)NS * 8));
    GM_TRY(segkalt.alloc_scratch((size_t)NS * 8));
    GM_TRY(segval.alloc_scratch((size_t)NS * 4));
    GM_TRY(segbin.alloc_scratch((size_t)NS * 8));
    hipLaunchKernelGGL(pb_segments_kernel, dim3(gm_), dim3(256), 0, 0, ckeys, flag.as<uint32_t>(),
                       segid.as<uint32_t>(), m, bin_bits, sb, pl->s_log, jb, vstart.as<uint32_t>(), segkey.as<uint64_t>(),
                       segbin.as<uint64_t>());
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize());
    flag.release();
    // first segment of every bin (segments are bin-major already)
    DevBuf bin_seg;
    GM_TRY(bin_seg.alloc(((size_t)Bv + 1) * 4));
    const unsigned gs = pb_grid((uint64_t)NS + 1);
    hipLaunchKernelGGL(pb_bounds_kernel, dim3(gs), dim3(256), 0, 0, segbin.as<uint64_t>(), NS, 0, Bv,
                       bin_seg.as<uint32_t>());
    GM_HIP(hipGetLastError());
    timer.done("pb plan: segments");
    // phase-1 order of the segments: by (tile, bin); the bin-major index rides in the low bits of the key (the same
    // keys-only radix sort as the edges: one instantiation of rocPRIM's sort in the code object instead of three)
    GM_TRY(sort_keys_u64(segkey, segkalt, NS, 0, tile_bits + tile_shift));
    segkalt.release();
    segbin.release();
    hipLaunchKernelGGL(pb_segval_kernel, dim3(gs), dim3(256), 0, 0, segkey.as<uint64_t>(), NS, jb, segval.as<uint32_t>());
    GM_HIP(hipGetLastError());

    DevBuf cnt, cs, cntv, vstart4, tile_seg, tile_pad, pstart, rank_of;
    GM_TRY(cnt.alloc_scratch(((size_t)NS + 1) * 4));
    GM_TRY(cs.alloc_scratch(((size_t)NS + 1) * 4));
    GM_TRY(cntv.alloc_scratch(((size_t)NS + 1) * 4));
    GM_TRY(vstart4.alloc_scratch(((size_t)NS + 1) * 4));
    GM_TRY(tile_seg.alloc(((size_t)pl->NT + 1) * 4));
    GM_TRY(tile_pad.alloc(((size_t)pl->NT + 1) * 4));
    GM_TRY(pstart.alloc_scratch((size_t)NS * 4));
    GM_TRY(rank_of.alloc_scratch((size_t)NS * 4));
    GM_TRY(pl->delta.alloc((size_t)NS * 4));
    hipLaunchKernelGGL(pb_seg_counts_kernel, dim3(gs), dim3(256), 0, 0, segval.as<uint32_t>(), vstart.as<uint32_t>(), NS,
                       m, cnt.as<uint32_t>(), cntv.as<uint32_t>(),
                       // GM_PB_SEGPAD=8 (measurement): every segment starts on a 32-byte sector of the value stream
                       pb_env_m("GM_PB_SEGPAD", (int)PB_VEC) == 8 ? 8u : PB_VEC);
    GM_HIP(hipGetLastError());
    GM_TRY(scan_exclusive<uint32_t>(cnt.as<uint32_t>(), cs.as<uint32_t>(), (uint64_t)NS + 1));
    hipLaunchKernelGGL(pb_bounds_kernel, dim3(gs), dim3(256), 0, 0, segkey.as<uint64_t>(), NS, tile_shift, pl->NT,
                       tile_seg.as<uint32_t>());
    hipLaunchKernelGGL(pb_tile_sizes_kernel, dim3(pb_grid((uint64_t)pl->NT + 1)), dim3(256), 0, 0,
                       tile_seg.as<uint32_t>(), cs.as<uint32_t>(), pl->NT, tile_pad.as<uint32_t>());
    hipLaunchKernelGGL(pb_tile_tail_kernel, dim3(pb_grid(pl->NT)), dim3(256), 0, 0, tile_seg.as<uint32_t>(),
                       cs.as<uint32_t>(), tile_pad.as<uint32_t>(), segval.as<uint32_t>(), pl->NT, cntv.as<uint32_t>());
    if (pb_env_m("GM_PB_BIN_GAP", 0) >= 8)
        hipLaunchKernelGGL(pb_bin_gap_kernel, dim3(pb_grid(Bv)), dim3(256), 0, 0, bin_seg.as<uint32_t>(), Bv,
                           (uint32_t)pb_env_m("GM_PB_BIN_GAP", 0), cntv.as<uint32_t>());
    GM_HIP(hipGetLastError());
    GM_TRY(scan_exclusive<uint32_t>(cntv.as<uint32_t>(), vstart4.as<uint32_t>(), (uint64_t)NS + 1));
    uint32_t Mv = 0;
    GM_HIP(hipMemcpy(&Mv, vstart4.as<uint32_t>() + NS, 4, hipMemcpyDeviceToHost));
    pl->Mv = Mv;
    hipLaunchKernelGGL(pb_bin_ranges_kernel, dim3(pb_grid((uint64_t)Bv + 1)), dim3(256), 0, 0, bin_seg.as<uint32_t>(),
                       vstart4.as<uint32_t>(), Bv, pl->bin_v.as<uint32_t>());
    GM_HIP(hipGetLastError());
    GM_TRY(scan_exclusive<uint32_t>(tile_pad.as<uint32_t>(), pl->tile_p.as<uint32_t>(), (uint64_t)pl->NT + 1));
    uint32_t Mp = 0;
    GM_HIP(hipMemcpy(&Mp, pl->tile_p.as<uint32_t>() + pl->NT, 4, hipMemcpyDeviceToHost));
    pl->Mp = Mp;
    hipLaunchKernelGGL(pb_seg_layout_kernel, dim3(gs), dim3(256), 0, 0, segkey.as<uint64_t>(), segval.as<uint32_t>(),
                       vstart4.as<uint32_t>(), cs.as<uint32_t>(), tile_seg.as<uint32_t>(), pl->tile_p.as<uint32_t>(), NS,
                       tile_shift, pstart.as<uint32_t>(), pl->delta.as<uint32_t>(), rank_of.as<uint32_t>());
    GM_HIP(hipGetLastError());

    if (pb_getenv_m("GM_PB_SPREAD_PLAN")) { // measurement: the two index streams spread the same way
        unsigned mib = 64, factor = 8, seed = 1;
        (void)sscanf(pb_getenv_m("GM_PB_SPREAD_PLAN"), "%u,%u,%u", &mib, &factor, &seed);
        GM_TRY(pl->p2_dst.alloc_spread((size_t)Mv * 2, (size_t)mib << 20, factor, seed + 101));
        GM_TRY(pl->p1_src.alloc_spread((size_t)Mp * 2, (size_t)mib << 20, factor, seed + 202));
    } else { // streamed once per sweep: pieces from all over the arena (arena.hip)
        GM_TRY(pl->p2_dst.alloc_big((size_t)Mv * 2, 0x9D57, 4, 0, ~0ull, 0, 4));
        GM_TRY(pl->p1_src.alloc_big((size_t)Mp * 2, 0x9157, 4, 0, ~0ull, 0, 4));
    }
    GM_TRY(pl->chunk_seg.alloc(((size_t)Mp / PB_WBLK + 1) * 4));
    GM_HIP(hipMemset(pl->p1_src.p, 0x7F, (size_t)Mp * 2)); // padding: an unflagged id (any source of the tile will do)
    GM_HIP(hipMemset(pl->p2_dst.p, 0xFF, (size_t)Mv * 2));
    // the sources of the hub groups' stream entries (the stream's last bins): what their blocks are merged by
    DevBuf hubsrc;
    uint32_t hub_q0 = Mv;
    if (pl->G) {
        GM_HIP(hipMemcpy(&hub_q0, pl->bin_v.as<uint32_t>() + pl->B, 4, hipMemcpyDeviceToHost));
        GM_TRY(hubsrc.alloc_scratch((size_t)(Mv - hub_q0 + 1) * 4));
        GM_HIP(hipMemset(hubsrc.p, 0xFF, (size_t)(Mv - hub_q0 + 1) * 4));
    }
    hipLaunchKernelGGL(pb_fill_kernel, dim3(gm_), dim3(256), 0, 0, ckeys, segid.as<uint32_t>(),
                       vstart.as<uint32_t>(), vstart4.as<uint32_t>(), rank_of.as<uint32_t>(), pstart.as<uint32_t>(), m,
                       bin_bits, sb, pl->s_log, pl->p1_src.as<uint16_t>(), pl->p2_dst.as<uint16_t>(),
                       pl->G ? hubsrc.as<uint32_t>() : (uint32_t *)nullptr, hub_q0);
    if (pl->G && Mv > hub_q0)
        hipLaunchKernelGGL(pb_hubsrc_pad_kernel, dim3(pb_grid(Mv - hub_q0)), dim3(256), 0, 0, hubsrc.as<uint32_t>(), Mv - hub_q0);
    hipLaunchKernelGGL(pb_chunk_seg_kernel, dim3(pb_grid(Mp / PB_WBLK + 1)), dim3(256), 0, 0, pstart.as<uint32_t>(), NS,
                       Mp / PB_WBLK + 1, pl->chunk_seg.as<uint32_t>());
    GM_HIP(hipGetLastError());
    timer.done("pb plan: segment layout + stream fill");
    GM_TRY(pb_make_items(pl));
    if (pl->G && pl->hub_csr) {
        // every hub row an item list of pb_hublong_kernel<true>: its stretch of hub_gidx, its group's part of the stream
        std::vector<uint32_t> g0(pl->n_hub), rq0(pl->n_hub), rq1(pl->n_hub);
        std::vector<PbHubItem> rows(pl->n_hub);
        uint64_t Mg = 0;
        for (const PbHubItem &grp : pl->hub_items_host)
            for (uint32_t j = 0; j < grp.nh; ++j) {
                const uint32_t h = grp.row0 + j;
                rq0[h] = grp.q0, rq1[h] = grp.q1;
            }
        for (uint32_t h = 0; h < pl->n_hub; ++h) {
            g0[h] = (uint32_t)Mg;
            rows[h] = PbHubItem{(uint32_t)Mg, (uint32_t)Mg + pl->hub_degs_host[h], 0u, h, pl->G + h};
            Mg += (pl->hub_degs_host[h] + 3u) & ~3u;
        }
        GM_CHECK(Mg < (1ull << 32), GM_ERR_RANGE, "pb_build: %llu hub terms in CSR order exceed the 32-bit index", (unsigned long long)Mg);
        std::stable_sort(rows.begin(), rows.end(), [](const PbHubItem &a, const PbHubItem &c) { return a.q1 - a.q0 > c.q1 - c.q0; });
        GM_TRY(pl->hub_gidx.alloc_big((size_t)(Mg ? Mg : 4) * 4, 0x61D7, 4, 0, ~0ull, 0, 4));
        GM_TRY(pl->csr_items.alloc(rows.size() * sizeof(PbHubItem)));
        GM_HIP(hipMemcpy(pl->csr_items.p, rows.data(), rows.size() * sizeof(PbHubItem), hipMemcpyHostToDevice));
        DevBuf d_g0, d_q0, d_q1, d_missing;
        GM_TRY(d_g0.alloc((size_t)pl->n_hub * 4));
        GM_TRY(d_q0.alloc((size_t)pl->n_hub * 4));
        GM_TRY(d_q1.alloc((size_t)pl->n_hub * 4));
        GM_TRY(d_missing.alloc(4));
        GM_HIP(hipMemcpy(d_g0.p, g0.data(), (size_t)pl->n_hub * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemcpy(d_q0.p, rq0.data(), (size_t)pl->n_hub * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemcpy(d_q1.p, rq1.data(), (size_t)pl->n_hub * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemset(d_missing.p, 0, 4));
        hipLaunchKernelGGL(pb_hubcsr_index_kernel, dim3(pl->n_hub < 65536u ? pl->n_hub : 65536u), dim3(256), 0, 0, csr->offsets,
                           csr->targets, pl->hub_rows.as<uint32_t>(), d_g0.as<uint32_t>(), d_q0.as<uint32_t>(), d_q1.as<uint32_t>(),
                           hubsrc.as<uint32_t>(), hub_q0, pl->n_hub, pl->hub_gidx.as<uint32_t>(), d_missing.as<uint32_t>());
        GM_HIP(hipGetLastError());
        uint32_t missing = 0;
        GM_HIP(hipMemcpy(&missing, d_missing.p, 4, hipMemcpyDeviceToHost));
        GM_CHECK(missing == 0, GM_ERR_INVALID, "pb_build: %u hub terms have no entry in their group's part of the stream", missing);
        pl->n_long_rows = pl->n_hub;
        GM_TRY(pb_make_long_items(pl, rows, pl->n_hub));
        timer.done("pb plan: hub rows in CSR order (%u rows, %llu index entries)", pl->n_hub, (unsigned long long)Mg);
    }
    if (pl->G > pl->G_long && !pl->hub_csr) { // the hub groups walked by pb_hubseq_kernel: blocks, places, hot records
        const uint32_t GS = pl->G - pl->G_long;
        // where each group's hot-hub keys lie (sorted by (virtual bin, source))
        std::vector<uint32_t> hk_start((size_t)Bv + 1, 0u);
        if (mhh) {
            DevBuf d_start;
            GM_TRY(d_start.alloc(((size_t)Bv + 1) * 4));
            hipLaunchKernelGGL(pb_bounds_kernel, dim3(pb_grid(mhh)), dim3(256), 0, 0, hk, mhh, sb, Bv,
                               d_start.as<uint32_t>(), (uint32_t)((1ull << bin_bits) - 1ull));
            GM_HIP(hipGetLastError());
            GM_HIP(hipMemcpy(hk_start.data(), d_start.p, ((size_t)Bv + 1) * 4, hipMemcpyDeviceToHost));
        }
        std::vector<PbSeqGroup> groups(GS);
        std::vector<uint32_t> sfirst(GS + 1, 0u);
        uint32_t ent = 0;
        for (uint32_t i = 0; i < GS; ++i) {
            const PbHubItem &it = pl->hub_items_host[pl->G_long + i];
            PbSeqGroup &gr = groups[i];
            gr.q0 = it.q0, gr.q1 = it.q1, gr.nh = it.nh;
            gr.hk0 = hk_start[pl->B + it.group], gr.hk1 = hk_start[pl->B + it.group + 1];
            gr.ent0 = ent;
            ent += gr.hk1 - gr.hk0;
            const uint64_t total = (uint64_t)(gr.q1 - gr.q0) + (gr.hk1 - gr.hk0);
            gr.blk0 = sfirst[i];
            gr.nblk = (uint32_t)((total + PB_SEQ_CAP - 1) / PB_SEQ_CAP);
            sfirst[i + 1] = sfirst[i] + gr.nblk;
        }
        GM_CHECK(ent == mhh, GM_ERR_INVALID, "pb_build: %u hot terms of hub groups, %u keys taken out", ent, mhh);
        pl->seq_blocks = sfirst.back();
        GM_TRY(pl->seq_blk_first.alloc(sfirst.size() * 4));
        GM_HIP(hipMemcpy(pl->seq_blk_first.p, sfirst.data(), sfirst.size() * 4, hipMemcpyHostToDevice));
        DevBuf d_groups;
        GM_TRY(d_groups.alloc(groups.size() * sizeof(PbSeqGroup)));
        GM_HIP(hipMemcpy(d_groups.p, groups.data(), groups.size() * sizeof(PbSeqGroup), hipMemcpyHostToDevice));
        GM_TRY(pl->seq_blk.alloc((size_t)(pl->seq_blocks ? pl->seq_blocks : 1) * sizeof(uint4)));
        GM_TRY(pl->seq_rows.alloc((size_t)(pl->seq_blocks ? pl->seq_blocks : 1) * PB_HUB_MAX * 4));
        GM_TRY(pl->hh_ent.alloc((size_t)(mhh ? mhh : 1) * 4));
        if (pl->seq_blocks) {
            const uint64_t smask = (1ull << sb) - 1ull;
            hipLaunchKernelGGL(pb_hubseq_blocks_kernel, dim3(div_up(pl->seq_blocks, 256)), dim3(256), 0, 0, d_groups.as<PbSeqGroup>(), GS,
                               pl->seq_blocks, hubsrc.as<uint32_t>(), hub_q0, hk, smask, pl->seq_blk.as<uint4>());
            hipLaunchKernelGGL(pb_hubseq_layout_kernel, dim3(pl->seq_blocks), dim3(PB_SEQ_STEP / PB_VEC), 0, 0, d_groups.as<PbSeqGroup>(), GS,
                               pl->seq_blk.as<uint4>(), hubsrc.as<uint32_t>(), hub_q0, hk, smask, sb + bin_bits + 1,
                               (uint32_t)((1u << rb) - 1u), hot_rank.as<uint32_t>(), pl->p2_dst.as<uint16_t>(),
                               pl->hh_ent.as<uint32_t>(), pl->seq_rows.as<uint32_t>());
            GM_HIP(hipGetLastError());
            GM_HIP(hipDeviceSynchronize());
        }
        timer.done("pb plan: hub groups row-major (%u blocks, %u hot records)", pl->seq_blocks, mhh);
    }
    hubsrc.release();
    hot_rank.release();
    // phase-1 workgroup list: a tile's stream is cut into chunks of 24576 entries (enough workgroups to hide latency, x-tile
    // reloads stay in L2).  Round 4, with the value stream on well-spread pages (tools/runs/r04_call21.sh / 22: fresh
    // processes on one box, ms per sweep at scale 26): 8192 2.99-3.01, 16384 2.68, 20480 2.61-2.62, 22528 2.61-2.63, 24576
    // 2.62-2.66, 28672 2.66-2.70, 32768 (the default until then) 2.63-2.71, 49152 2.73-2.76; scale 24 0.663 against 0.673,
    // scale 22 unchanged
    {
        uint64_t chunk = 24576;
        if (pb_env("GM_PB_CHUNK", 0) > 0)
            chunk = ((uint64_t)pb_env("GM_PB_CHUNK", 0) + PB_WBLK - 1) & ~(uint64_t)(PB_WBLK - 1);
        pl->chunk = (uint32_t)chunk;
    }
    DevBuf wg_cnt, wg_first;
    GM_TRY(wg_cnt.alloc(((size_t)pl->NT + 1) * 4));
    GM_TRY(wg_first.alloc(((size_t)pl->NT + 1) * 4));
    hipLaunchKernelGGL(pb_wg_count_kernel, dim3(pb_grid((uint64_t)pl->NT + 1)), dim3(256), 0, 0, pl->tile_p.as<uint32_t>(),
                       pl->NT, pl->chunk, wg_cnt.as<uint32_t>());
    GM_HIP(hipGetLastError());
    GM_TRY(scan_exclusive<uint32_t>(wg_cnt.as<uint32_t>(), wg_first.as<uint32_t>(), (uint64_t)pl->NT + 1));
    uint32_t NW = 0;
    GM_HIP(hipMemcpy(&NW, wg_first.as<uint32_t>() + pl->NT, 4, hipMemcpyDeviceToHost));
    pl->NW = NW;
    GM_TRY(pl->wg_tile.alloc((size_t)NW * 4));
    GM_TRY(pl->wg_p0.alloc((size_t)NW * 4));
    hipLaunchKernelGGL(pb_wg_fill_kernel, dim3(pb_grid(NW)), dim3(256), 0, 0, pl->tile_p.as<uint32_t>(),
                       wg_first.as<uint32_t>(), pl->NT, pl->chunk, NW, pl->wg_tile.as<uint32_t>(), pl->wg_p0.as<uint32_t>());
    pl->xcd_aware = pb_env_m("GM_PB_XCD", 1);
    pl->wg_first_host.resize((size_t)pl->NT + 1);
    GM_HIP(hipMemcpy(pl->wg_first_host.data(), wg_first.p, ((size_t)pl->NT + 1) * 4, hipMemcpyDeviceToHost));
    GM_HIP(hipGetLastError());
    GM_HIP(hipDeviceSynchronize());
    if (const uint32_t G = (uint32_t)pb_env_m("GM_PB_WG_GROUP", 0)) {
        // Tile-major order has the chunks of one tile next to each other: the workgroups running at one time on an XCD
        // are a few tiles x all their chunks, and what they write to one bin is a few adjacent runs.  Group-major order
        // (G tiles x ONE chunk at a time) makes it G adjacent runs per bin — longer contiguous stretches of the value
        // stream filled at one time — at the price of re-reading a tile's out_scores one group pass later.
        std::vector<uint32_t> tiles(NW), p0s(NW), tg, pg;
        GM_HIP(hipMemcpy(tiles.data(), pl->wg_tile.p, (size_t)NW * 4, hipMemcpyDeviceToHost));
        GM_HIP(hipMemcpy(p0s.data(), pl->wg_p0.p, (size_t)NW * 4, hipMemcpyDeviceToHost));
        tg.reserve(NW), pg.reserve(NW);
        for (uint32_t t0 = 0; t0 < pl->NT; t0 += G) {
            const uint32_t t1 = t0 + G < pl->NT ? t0 + G : pl->NT;
            for (uint32_t c = 0;; ++c) {
                bool any = false;
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint32_t w = pl->wg_first_host[t] + c;
                    if (w < pl->wg_first_host[t + 1]) {
                        tg.push_back(tiles[w]), pg.push_back(p0s[w]);
                        any = true;
                    }
                }
                if (!any)
                    break;
            }
        }
        GM_CHECK(tg.size() == NW, GM_ERR_INVALID, "pb_build: grouped work list has %zu of %u items", tg.size(), NW);
        GM_TRY(pl->wg_tile_g.alloc((size_t)NW * 4));
        GM_TRY(pl->wg_p0_g.alloc((size_t)NW * 4));
        GM_HIP(hipMemcpy(pl->wg_tile_g.p, tg.data(), (size_t)NW * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemcpy(pl->wg_p0_g.p, pg.data(), (size_t)NW * 4, hipMemcpyHostToDevice));
    }
    return GM_OK;
}

} // namespace

static hipError_t pb_set_kernel_attributes()
{
    const void *bin_fns[] = {
        reinterpret_cast<const void *>(&pb_bin_kernel<0, 14>), reinterpret_cast<const void *>(&pb_bin_kernel<1, 14>),
        reinterpret_cast<const void *>(&pb_bin_kernel<3, 14>), reinterpret_cast<const void *>(&pb_bin_kernel<4, 14>),
        reinterpret_cast<const void *>(&pb_bin_kernel<5, 14>), reinterpret_cast<const void *>(&pb_bin_kernel<0, 15>),
        reinterpret_cast<const void *>(&pb_bin_kernel<3, 15>), reinterpret_cast<const void *>(&pb_bin_kernel<5, 15>)};
    const void *acc_fns[] = {reinterpret_cast<const void *>(&pb_accum_kernel<0>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<0, 2>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<0, 6>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<0, PB_ACC_DEPTH, true>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<0, PB_ACC_DEPTH, false, true>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<5, PB_ACC_DEPTH, false, true>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<6, PB_ACC_DEPTH, false, true>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<3>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<4>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<5>),
                             reinterpret_cast<const void *>(&pb_accum_kernel<6>)};
    hipError_t e = hipSuccess;
    for (const void *f : bin_fns)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (4u << PB_S_LOG_MAX) + PB_DCACHE * 4);
    for (const void *f : acc_fns)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 163840 - PB_ACC_STATIC);
    return e;
}

static bool pb_bin_dispatch(const PbPlan *pl, PbScratch *sc, const float *x_in, uint32_t w_first, uint32_t w_count,
                            hipStream_t st, const uint32_t *item_list = nullptr, bool fold_hot = false, hipEvent_t stop = nullptr);

int pb_plan_create(const gm_csr *csr, uint64_t x_len, PbPlan **out)
{
    PbPlan *pl = new (std::nothrow) PbPlan();
    GM_CHECK(pl, GM_ERR_NOMEM, "pb_plan_create: out of host memory");
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = pb_build(csr, x_len, pl);
    pl->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != GM_OK) {
        delete pl;
        return rc;
    }
    hipError_t e = pb_set_kernel_attributes();
    if (e != hipSuccess) {
        set_error("pb_plan_create: %s", hipGetErrorString(e));
        delete pl;
        return GM_ERR_HIP;
    }
    *out = pl;
    return GM_OK;
}

// The plan depends only on the CSR and on x_len: build it on first use and keep it in the handle, so
// repeated page_rank() calls on one graph (the reference's app: warm-up + timed runs) pay for it once.
// Plans are shared: an engine keeps its plan alive for as long as it lives, whatever happens to the cache.
// GM_PB_NOCACHE=1 (measurement tools that switch plan knobs on one resident graph) builds a private plan
// for the caller and leaves the cached one alone.
int pb_plan_get(const gm_csr *csr, uint64_t x_len, std::shared_ptr<const PbPlan> *out)
{
    if (pb_env("GM_PB_NOCACHE", 0)) {
        PbPlan *pl = nullptr;
        GM_TRY(pb_plan_create(csr, x_len, &pl));
        out->reset(pl, [](const PbPlan *p) { delete p; });
        return GM_OK;
    }
    std::lock_guard<std::mutex> lock(csr->cache_mu);
    auto it = csr->pb_plans.find(x_len);
    if (it == csr->pb_plans.end()) {
        PbPlan *pl = nullptr;
        GM_TRY(pb_plan_create(csr, x_len, &pl));
        it = csr->pb_plans.emplace(x_len, std::shared_ptr<const PbPlan>(pl, [](const PbPlan *p) { delete p; })).first;
    }
    *out = it->second;
    return GM_OK;
}


int pb_scratch_create(const PbPlan *pl, PbScratch **out, DevBuf *early)
{
    PbScratch *sc = new (std::nothrow) PbScratch();
    GM_CHECK(sc, GM_ERR_NOMEM, "pb_scratch_create: out of host memory");
    int rc;
    if (early && early->p && early->bytes >= (size_t)(pl->Mv ? pl->Mv : 4) * 4)
        sc->vals_raw = std::move(*early);
#ifdef GM_MEASURE // the measurement library only: other ways of giving the value stream its memory (round 3's placement study)
    // GM_PB_VALS_SLACK=<MiB> (measurements): room behind the value stream so that GM_PB_VALS_OFFSET=<KiB>, read at
    // every sweep, can move it inside one allocation — does the sweep time depend on the offset or on the pages?
    const size_t slack = (size_t)pb_env("GM_PB_VALS_SLACK", 0) << 20;
    if (!sc->vals_raw.p && pb_env("GM_PB_VALS_CONTIG", 0)) { // measurement: physically contiguous pages
        void *p = nullptr;
        const size_t bytes = (size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            sc->vals_raw.p = p;
            sc->vals_raw.bytes = bytes;
        } else {
            (void)hipGetLastError();
        }
    }
    // GM_PB_VALS_VMM=<MiB per physical piece, -1: one piece> (measurement): the stream as a virtual range mapped from
    // separately allocated physical pieces (HIP virtual-memory API); GM_PB_VALS_VMM_ALIGN=<MiB> aligns the range,
    // GM_PB_VALS_VMM_SHUFFLE=<seed> maps the pieces in a pseudo-random order
    // GM_PB_VALS_SHARE=<draw id> (measurement): engines created with the same id use ONE allocation (kept for the life of
    // the process; one engine at a time may sweep) — variants of the kernels compared on identical pages
    if (!sc->vals_raw.p && pb_env("GM_PB_VALS_SHARE", 0)) {
        static std::mutex mu;
        static std::map<int, std::shared_ptr<DevBuf>> draws;
        std::lock_guard<std::mutex> lock(mu);
        std::shared_ptr<DevBuf> &d = draws[pb_env("GM_PB_VALS_SHARE", 0)];
        const size_t bytes = ((size_t)pl->m + ((size_t)pl->m >> 3) + (64u << 20)) * 4; // room for every variant's padding
        if (!d) {
            d = std::make_shared<DevBuf>();
            if ((rc = pb_env("GM_PB_VALS_VMM", 0) ? d->alloc_vmm(bytes, pb_env("GM_PB_VALS_VMM", 0) < 0 ? 0 : (size_t)pb_env("GM_PB_VALS_VMM", 0) << 20, 0)
                                                  : d->alloc(bytes))) {
                d.reset();
                delete sc;
                return rc;
            }
        }
        if (d->bytes >= (size_t)(pl->Mv ? pl->Mv : 4) * 4) {
            sc->vals_shared = d;
            sc->vals_raw.p = d->p; // borrowed: detached again in the destructor
            sc->vals_raw.bytes = d->bytes;
        }
    }
    // GM_PB_VALS_POOL="<MiB per piece>,<pieces created>,<first>,<stride>" (measurement): the stream mapped from every
    // stride-th of a sequence of physical pieces created back to back (DevBuf::alloc_vmm_strided)
    if (!sc->vals_raw.p && getenv("GM_PB_VALS_POOL")) {
        unsigned mib = 256, pool = 0, first = 0, stride = 1;
        (void)sscanf(getenv("GM_PB_VALS_POOL"), "%u,%u,%u,%u", &mib, &pool, &first, &stride);
        if ((rc = sc->vals_raw.alloc_vmm_strided((size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack, (size_t)mib << 20, pool, first, stride))) {
            delete sc;
            return rc;
        }
    }
    // GM_PB_VALS_PICK="<MiB per piece>,<pieces created>,i0,i1,..." (measurement): the stream mapped from exactly the pieces
    // i0, i1, ... of a sequence of physical pieces created back to back (which stretches of physical memory go together?)
    if (!sc->vals_raw.p && getenv("GM_PB_VALS_PICK")) {
        std::vector<size_t> nums;
        for (const char *q = getenv("GM_PB_VALS_PICK"); *q;) {
            nums.push_back((size_t)strtoull(q, const_cast<char **>(&q), 10));
            while (*q == ',')
                ++q;
        }
        if (nums.size() >= 3) {
            const size_t chunk = nums[0] << 20, pool = nums[1];
            std::vector<size_t> pick(nums.begin() + 2, nums.end());
            const size_t bytes = (size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack;
            while (pick.size() * chunk < bytes)
                pick.push_back(pick.back() + 1);
            if ((rc = sc->vals_raw.alloc_vmm_pool(bytes, chunk, pool, pick))) {
                delete sc;
                return rc;
            }
        }
    }
    // GM_PB_SPREAD="<MiB per piece>,<pool factor>,<seed>": a pseudo-random subset of a pool of pieces (DevBuf::alloc_spread)
    if (!sc->vals_raw.p && pb_getenv_m("GM_PB_SPREAD")) {
        unsigned mib = 64, factor = 8, seed = 1;
        (void)sscanf(pb_getenv_m("GM_PB_SPREAD"), "%u,%u,%u", &mib, &factor, &seed);
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = sc->vals_raw.alloc_spread((size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack, (size_t)mib << 20, factor, seed))) {
            delete sc;
            return rc;
        }
        if (log_enabled())
            fprintf(stderr, "[graph_mi355x] value stream spread over %zu pieces of %u MiB (pool x%u) took %.2f ms\n",
                    sc->vals_raw.vmm.size(), mib, factor,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    if (!sc->vals_raw.p && pb_env("GM_PB_VALS_VMM", 0)) {
        const size_t bytes = (size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack;
        const int mib = pb_env("GM_PB_VALS_VMM", 0);
        const size_t chunk = mib < 0 ? 0 : (size_t)mib << 20;
        std::vector<uint32_t> order;
        if (chunk && pb_env("GM_PB_VALS_VMM_SHUFFLE", 0)) {
            const size_t gran = 2u << 20, c = (chunk + gran - 1) / gran * gran, count = (bytes + c - 1) / c;
            order.resize(count);
            for (size_t i = 0; i < count; ++i)
                order[i] = (uint32_t)i;
            uint64_t state = (uint64_t)pb_env("GM_PB_VALS_VMM_SHUFFLE", 0) * 0x9E3779B97F4A7C15ull + 1;
            for (size_t i = count; i > 1; --i) { // Fisher-Yates with a splitmix-style generator
                state += 0x9E3779B97F4A7C15ull;
                uint64_t z = state;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                std::swap(order[i - 1], order[z % i]);
            }
        }
        if ((rc = sc->vals_raw.alloc_vmm(bytes, chunk, (size_t)pb_env("GM_PB_VALS_VMM_ALIGN", 0) << 20,
                                         order.empty() ? nullptr : order.data()))) {
            delete sc;
            return rc;
        }
    }
#else
    const size_t slack = 0;
#endif
    // The default: the stream mapped from 64 MiB pieces of the arena (arena.hip — right after a plan build its free list
    // holds the build's ~20-30 GB of temporaries), several candidate sets timed with the bin kernel itself, the fastest
    // kept.  Which physical memory the stream lies in — relative to the index stream read beside it — decides up to a third
    // of the bin kernel's time (1.11-1.22 ms against 1.41-1.59 at RMAT scale 26, profiles/r03_placement_*.txt); a candidate
    // costs a remap and three launches (~5 ms).  GM_PB_DRAWS (4) candidates of different kinds are tried until one reaches
    // GM_PB_BW_OK (4000 GB/s); if the best stays below GM_PB_BW_MIN (3700 GB/s: between the medium and the slow level) the
    // arena is grown by 8 GiB at a time (GM_PB_GROW_GIB in all, default 16) and every fresh stretch tried alone and mixed
    // with the pool.  What this search cannot reach (tools/runs/r03_call45.sh, profiles/r03_placement_search_log.txt): in about
    // one process in four the bin kernel runs at 1.36-1.52 ms whatever the value stream is mapped from — ten candidates of
    // all kinds within 1 % of each other — and re-placing the plan's index stream (copied into the pool's oldest / newest
    // pieces, spread sets, fresh memory) does not move it either; a second plan of the same graph in the same process may
    // run at another level.  That state belongs to the process (or the plan's small arrays), not to these two streams.
    if (!sc->vals_raw.p && arena_enabled() && (size_t)pl->Mv * 4 + slack >= ARENA_MIN && pl->NW && pb_env("GM_PB_DRAWS", 4) > 0) {
        const size_t bytes = (size_t)pl->Mv * 4 + slack;
        const int draws = pb_env("GM_PB_DRAWS", 4);
        const double moved = (double)pl->Mp * 2 + (double)pl->Mv * 4 + (double)pl->x_len * 4; // bytes of one bin launch
        const double bw_min = (double)pb_env("GM_PB_BW_MIN", 3700) * 1e9;
        DevBuf probe_x; // any readable x will do for the timing
        hipEvent_t e0 = nullptr, e1 = nullptr;
        float best_ms = 0.f;
        rc = draws > 1 ? probe_x.alloc((size_t)pl->x_len * 4) : GM_OK;
        hipError_t he = hipSuccess;
        if (rc == GM_OK && draws > 1) {
            he = hipMemset(probe_x.p, 0, (size_t)pl->x_len * 4);
            if (he == hipSuccess)
                he = hipEventCreate(&e0);
            if (he == hipSuccess)
                he = hipEventCreate(&e1);
        }
        int tried = 0;
        // times one candidate and keeps it if it is the fastest so far (the loser goes back to the arena)
        auto consider = [&](DevBuf &cand, const char *what) {
            sc->vals = cand.as<float>();
            pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0); // clocks up, pages touched
            he = hipEventRecord(e0, (hipStream_t)0);
            pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0);
            pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0);
            if (he == hipSuccess)
                he = hipEventRecord(e1, (hipStream_t)0);
            if (he == hipSuccess)
                he = hipEventSynchronize(e1);
            float ms = 0.f;
            if (he == hipSuccess)
                he = hipEventElapsedTime(&ms, e0, e1);
            ms *= 0.5f;
            if (log_enabled())
                fprintf(stderr, "[graph_mi355x] value stream draw %d (%s): bin kernel %.3f ms = %.0f GB/s\n", tried, what, ms,
                        moved / (ms * 1e-3) / 1e9);
            ++tried;
            if (he == hipSuccess) {
                const uint32_t us = (uint32_t)(ms * 1000.0f);
                sc->draw_best_us = sc->draws_timed == 0 || us < sc->draw_best_us ? us : sc->draw_best_us;
                sc->draw_worst_us = us > sc->draw_worst_us ? us : sc->draw_worst_us;
                ++sc->draws_timed;
            }
            if (he == hipSuccess && (!sc->vals_raw.p || ms < best_ms)) {
                best_ms = ms;
                sc->vals_raw = std::move(cand);
            }
            sc->vals = nullptr;
        };
        const size_t count = (bytes + ARENA_PIECE - 1) / ARENA_PIECE;
        const size_t step = ((size_t)8 << 30) / ARENA_PIECE > 2 * count ? ((size_t)8 << 30) / ARENA_PIECE : 2 * count;
        // fast enough to stop looking (GM_PB_BW_OK GB/s by the `moved` model: the fast level is 4060-4500, the slow one <= 3510)
        const double bw_ok = (double)pb_env("GM_PB_BW_OK", 4000) * 1e9;
        auto fast_enough = [&] { return sc->vals_raw.p && best_ms > 0.f && moved / (best_ms * 1e-3) >= bw_ok; };
        // Candidates that differ in KIND, not only in seed (three spread draws of one pool measure the same to 1 %): spread
        // over the whole pool; the pool's oldest pieces only; its newest only; spread again.  What makes a set of pieces fast
        // is not understood beyond the maps in profiles/r03_placement_*.txt — pieces that are fast alone can be slow mixed
        // and the other way round — so the stream is chosen by measurement.
        uint64_t pool_lo = 0, pool_hi = 0;
        size_t pool_free = 0;
        for (int k = 0; k < draws && rc == GM_OK && he == hipSuccess && !fast_enough(); ++k) {
            DevBuf cand;
            const char *what = "all over the arena";
            if (k == 0 || k >= 3 || draws == 1) {
                rc = cand.alloc_big(bytes, 0xA11CE5 + 7919ull * (uint64_t)k + pl->NS, 4, 0, ~0ull, 0, 8);
                arena_free_range(pl->device, &pool_lo, &pool_hi, &pool_free); // the pool as the first draw left it
            } else {
                const uint64_t window = 2 * count; // serials; pieces of the range that are in use elsewhere are simply missing
                const bool oldest = k == 1;
                what = oldest ? "the pool's oldest pieces" : "the pool's newest pieces";
                if (pool_hi - pool_lo < 3 * window || pool_free < 3 * window)
                    continue; // a pool this small has no distinct ends
                // (straight from the arena: a window without enough free pieces is not a candidate, no hipMalloc stand-in)
                const int rcw = oldest ? cand.alloc_from_arena(bytes, 0x01DE57 + pl->NS, 1, pool_lo, pool_lo + window, 0)
                                       : cand.alloc_from_arena(bytes, 0x0E3E57 + pl->NS, 1, pool_hi - window, pool_hi, 0);
                if (rcw != GM_OK)
                    continue;
            }
            if (rc != GM_OK || draws == 1) {
                if (rc == GM_OK)
                    sc->vals_raw = std::move(cand);
                break;
            }
            consider(cand, what);
        }
        // ... and if none of them is above the slow level, stretches of fresh memory behind the pool: alone, and half-and-half
        // with the pool; one candidate at a time (the loser's pieces and address range go back before the next is drawn)
        size_t budget = ((size_t)pb_env("GM_PB_GROW_GIB", 16) << 30) / ARENA_PIECE;
        while (draws > 1 && rc == GM_OK && he == hipSuccess && bytes >= ((size_t)1 << 30) && budget >= step && sc->vals_raw.p &&
               moved / (best_ms * 1e-3) < bw_min) {
            budget -= step;
            uint64_t first = 0;
            rc = arena_grow(pl->device, step, &first);
            if (rc != GM_OK) { // out of memory for the experiment: keep what we have
                rc = GM_OK;
                (void)hipGetLastError();
                break;
            }
            sc->grown_pieces += (uint32_t)step;
            for (int variant = 0; variant < 2 && he == hipSuccess && moved / (best_ms * 1e-3) < bw_min; ++variant) {
                DevBuf cand;
                const int rcc = variant == 0 ? cand.alloc_from_arena(bytes, 0xB0B + first, 1, first, first + step, 0)
                                             : cand.alloc_from_arena(bytes, 0xC0C + first, 1, pool_lo, first + step, first, pool_hi);
                if (rcc == GM_OK)
                    consider(cand, variant == 0 ? "a fresh 8 GiB behind the pool" : "half pool, half fresh");
            }
        }
        // Last resort.  In a process where every set of arena pieces is slow (about one in four: 1.45-1.59 ms), memory obtained
        // in OTHER ways was measured faster — fresh VMM pieces of 256 MiB 1.33 ms, plain hipMalloc 1.37 ms against 1.575 ms for
        // eight arena candidates (profiles/r03_placement_piece_size.txt, third box) — although both are slower than the
        // arena wherever the arena is fast.  So they are tried only here, and only kept if they win.
        if (draws > 1 && rc == GM_OK && he == hipSuccess && sc->vals_raw.p && moved / (best_ms * 1e-3) < bw_min &&
            pb_env_m("GM_PB_LAST_RESORT", 1)) {
            {
                DevBuf cand;
                if (cand.alloc_vmm(bytes, (size_t)256 << 20, 0) == GM_OK)
                    consider(cand, "fresh VMM pieces of 256 MiB");
                else
                    (void)hipGetLastError();
            }
            if (he == hipSuccess && moved / (best_ms * 1e-3) < bw_min) {
                DevBuf cand;
                if (cand.alloc(bytes) == GM_OK)
                    consider(cand, "plain hipMalloc");
                else
                    (void)hipGetLastError();
            }
        }
        if (const char *gm = pb_getenv_m("GM_PB_GROW_MAP")) { // measurement (profiles/r03_placement_grow_map.txt): which later stretches
            // of memory are fast alone, which pair well with the pool?  k-th stretch of 8 GiB created behind the pool, timed
            // alone and half-and-half with the pool as it was; one candidate at a time
            uint64_t pool_end = 0;
            (void)arena_grow(pl->device, 0, &pool_end);
            for (int k = 0; k < atoi(gm) && rc == GM_OK && he == hipSuccess && draws > 1; ++k) {
                uint64_t first = 0;
                if (arena_grow(pl->device, step, &first) != GM_OK)
                    break;
                float t[2] = {0.f, 0.f};
                for (int variant = 0; variant < 2; ++variant) {
                    DevBuf cand;
                    const int rcc = variant == 0 ? cand.alloc_from_arena(bytes, 0xE0E + first, 1, first, first + step, 0)
                                                 : cand.alloc_from_arena(bytes, 0xF0F + first, 1, 0, first + step, first, pool_end);
                    if (rcc != GM_OK)
                        continue;
                    sc->vals = cand.as<float>();
                    pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0);
                    he = hipEventRecord(e0, (hipStream_t)0);
                    pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0);
                    pb_bin_dispatch(pl, sc, probe_x.as<float>(), 0, pl->NW, (hipStream_t)0);
                    if (he == hipSuccess)
                        he = hipEventRecord(e1, (hipStream_t)0);
                    if (he == hipSuccess)
                        he = hipEventSynchronize(e1);
                    if (he == hipSuccess)
                        he = hipEventElapsedTime(&t[variant], e0, e1);
                    sc->vals = nullptr;
                }
                fprintf(stderr, "[graph_mi355x] grow map: stretch %2d (pieces %llu..%llu, pool ended at %llu): alone %.3f ms, half pool + half stretch %.3f ms\n",
                        k, (unsigned long long)first, (unsigned long long)(first + step), (unsigned long long)pool_end, t[0] * 0.5f, t[1] * 0.5f);
            }
        }
        sc->vals = nullptr;
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
        if (he != hipSuccess && rc == GM_OK) {
            set_error("pb_scratch_create: timing the value stream draws: %s", hipGetErrorString(he));
            rc = GM_ERR_HIP;
        }
        if (rc != GM_OK) {
            delete sc;
            return rc;
        }
    }
    if ((rc = sc->vals_raw.p ? GM_OK : sc->vals_raw.alloc((size_t)(pl->Mv ? pl->Mv : 4) * 4 + slack)) ||
        (rc = sc->partials.alloc((size_t)(pl->slots ? pl->slots : 1) * pl->Racc * 8)) ||
        (rc = sc->tickets.alloc((size_t)pl->B * 4)) || (rc = sc->bin_err.alloc(((size_t)pl->B + pl->err_slots) * 8)) ||
        (rc = sc->hot_x.alloc(((size_t)pl->H * pl->T + 4) * 4)) ||
        (rc = sc->long_state.alloc(8 + (size_t)pl->n_long_items * 8 + (size_t)pl->long_sbs_len * 4 + 8))) {
        delete sc;
        return rc;
    }
    sc->vals = sc->vals_raw.as<float>();
    {
        const int rc = sc->fold_ctr.alloc(16);
        if (rc != GM_OK) {
            delete sc;
            return rc;
        }
    }
    hipError_t e = hipMemset(sc->tickets.p, 0, (size_t)pl->B * 4);
    if (e == hipSuccess)
        e = hipMemset(sc->fold_ctr.p, 0, 16);
    if (e == hipSuccess) // (no hand-off word carries epoch 0, the counter starts at 0, sums of 0 predict nothing)
        e = hipMemset(sc->long_state.p, 0, sc->long_state.bytes);
    if (e == hipSuccess) // (an error slot no kernel of this plan writes — a hub row's or a hub group's, whichever sums it — stays 0)
        e = hipMemset(sc->bin_err.p, 0, sc->bin_err.bytes);
    if (e == hipSuccess)
        e = hipMemset(sc->vals_raw.p, 0, sc->vals_raw.bytes);
    // The hub kernels' streams get the LOWEST priority the device offers (GM_PB_SIDE_PRIO=0: the default one): their small
    // workgroups are meant to fill the room the accumulate workgroups leave on a CU, not to take CUs from them — several of
    // them on one CU leave no room for an accumulate workgroup (143 KiB of LDS) until they have finished.
    int prio_least = 0, prio_greatest = 0;
    if (e == hipSuccess && pb_env_m("GM_PB_SIDE_PRIO", 1))
        e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (pb_env_m("GM_PB_SIDE_PRIO", 1) == 2) // (measurements: the HIGHEST one)
        prio_least = prio_greatest;
    if (e == hipSuccess && pl->G_long && pl->G > pl->G_long) { // the long rows' own stream
        e = hipStreamCreateWithPriority(&sc->chain, hipStreamNonBlocking, prio_least);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&sc->ev_chain_fork, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&sc->ev_chain_join, hipEventDisableTiming);
    }
    if (e == hipSuccess && pl->G) { // the hub groups' own stream
        e = hipStreamCreateWithPriority(&sc->side, hipStreamNonBlocking, prio_least);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming);
    }
    // hipMemset on device memory returns before the fill has run, and the sweeps run on the caller's
    // (possibly non-blocking) stream, which the null stream does not order against
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("pb_scratch_create: %s", hipGetErrorString(e));
        delete sc;
        return GM_ERR_HIP;
    }
    *out = sc;
    return GM_OK;
}

void pb_scratch_destroy(PbScratch *scratch) { delete scratch; }

uint64_t pb_work_items(const PbPlan *plan) { return plan ? (uint64_t)plan->NW + plan->NI : 0; }

// diagnostics (gm_pr_plan_info): what the plan costs and what it contains
void pb_plan_info(const PbPlan *pl, const PbScratch *sc, uint64_t *info, uint32_t count)
{
    const DevBuf *bufs[] = {&pl->cidx, &pl->hub_rows, &pl->p1_src, &pl->chunk_seg, &pl->delta, &pl->tile_p, &pl->wg_tile,
                            &pl->wg_p0,  &pl->p2_dst, &pl->bin_v,  &pl->items,     &pl->hot_ids, &pl->hot_ent, &pl->hbin_v, &pl->hot_base,
                            &pl->seq_rows, &pl->seq_blk, &pl->hh_ent, &pl->hub_gidx, &pl->csr_items, &pl->long_items, &pl->long_rows};
    uint64_t plan_bytes = 0;
    for (const DevBuf *b : bufs)
        plan_bytes += b->bytes;
    const uint64_t scratch_bytes = sc ? sc->vals_raw.bytes + sc->partials.bytes + sc->tickets.bytes + sc->bin_err.bytes +
                                            sc->hot_x.bytes + sc->long_state.bytes : 0;
    const uint64_t v[] = {plan_bytes, (uint64_t)(pl->build_ms * 1000.0), pl->n_hub, pl->hub_edges, pl->hub_deg, pl->Htot,
                          pl->Mv, pl->Mh, scratch_bytes, pl->B, pl->NT, pl->NS, pl->G, pl->T, pl->n_long_rows, pl->long_terms, pl->seq_blocks,
                          sc ? sc->draw_best_us : 0u, sc ? sc->draw_worst_us : 0u, sc ? sc->draws_timed : 0u,
                          sc ? sc->grown_pieces : 0u, sc && !sc->vals_raw.arena.empty() ? 1u : 0u, pl->Mhh};
    for (uint32_t i = 0; i < count; ++i)
        info[i] = i < sizeof(v) / sizeof(v[0]) ? v[i] : 0;
}

// A launch whose packet does not carry the barrier bit (hipExtAnyOrderLaunch): it may start while the launches in front of
// it ON THE SAME STREAM are still running; the next ordinary launch of the stream waits for all of them.  That is the fork
// and the join of the accumulate phase (accumulate kernel beside the two hub kernels) without a second stream: the events
// that carried them cost 10-20 us each way (profiles/r04_*timeline*), a tenth of a sweep at scale 22.
template <typename... P, typename... A>
static hipError_t pb_launch_flags(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, bool any_order, A... a)
{
    static_assert(sizeof...(P) == sizeof...(A), "argument count");
    std::tuple<P...> vals(static_cast<P>(a)...);
    return std::apply(
        [&](auto &...v) {
            void *ptrs[] = {(void *)&v...};
            return hipExtLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, lds, st, nullptr, nullptr,
                                      any_order ? hipExtAnyOrderLaunch : 0);
        },
        vals);
}

// a launch whose COMPLETION is the event `stop` (hipExtLaunchKernel binds the event to the dispatch itself: no marker packet
// behind the kernel, which a hipEventRecord is — 6 us between the bin and the accumulate kernel at scale 22)
template <typename... P, typename... A>
static hipError_t pb_launch_stop(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t stop, A... a)
{
    static_assert(sizeof...(P) == sizeof...(A), "argument count");
    std::tuple<P...> vals(static_cast<P>(a)...);
    return std::apply(
        [&](auto &...v) {
            void *ptrs[] = {(void *)&v...};
            return hipExtLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, lds, st, nullptr, stop, 0);
        },
        vals);
}

template <int ABL, int S_LOG>
void pb_launch_bin(const PbPlan *pl, PbScratch *sc, const float *x_in, uint32_t w_first, uint32_t w_count, hipStream_t st,
                   const uint32_t *item_list = nullptr, bool fold_hot = false, hipEvent_t stop = nullptr)
{
    // the hot sources' share per workgroup must fit its threads: Htot <= w_count x 1024 on every whole sweep that has a
    // value stream worth the name; otherwise (and for partial launches) pb_hot_gather_kernel does it
    fold_hot = fold_hot && pl->Htot && (uint64_t)w_count * PB_BIN_BLOCK >= pl->Htot;
    const bool grouped = pl->wg_tile_g.p && w_first == 0 && w_count == pl->NW && !item_list; // whole sweeps only
    if (stop) {
        (void)pb_launch_stop(pb_bin_kernel<ABL, S_LOG>, dim3(w_count), dim3(PB_BIN_BLOCK), (4u << S_LOG) + PB_DCACHE * 4, st, stop, x_in,
                             pl->x_len, pl->tile_p.as<uint32_t>(), (grouped ? pl->wg_tile_g : pl->wg_tile).as<uint32_t>(),
                             (grouped ? pl->wg_p0_g : pl->wg_p0).as<uint32_t>(), pl->p1_src.as<uint16_t>(),
                             pl->chunk_seg.as<uint32_t>(), pl->delta.as<uint32_t>(), sc->vals, pl->chunk, w_first, pl->xcd_aware,
                             item_list, fold_hot ? pl->hot_ids.as<uint32_t>() : (const uint32_t *)nullptr, pl->Htot,
                             sc->hot_x.as<float>());
        return;
    }
    hipLaunchKernelGGL((pb_bin_kernel<ABL, S_LOG>), dim3(w_count), dim3(PB_BIN_BLOCK), (4u << S_LOG) + PB_DCACHE * 4, st, x_in,
                       pl->x_len, pl->tile_p.as<uint32_t>(), (grouped ? pl->wg_tile_g : pl->wg_tile).as<uint32_t>(),
                       (grouped ? pl->wg_p0_g : pl->wg_p0).as<uint32_t>(),
                       pl->p1_src.as<uint16_t>(), pl->chunk_seg.as<uint32_t>(), pl->delta.as<uint32_t>(), sc->vals,
                       pl->chunk, w_first, pl->xcd_aware, item_list, fold_hot ? pl->hot_ids.as<uint32_t>() : (const uint32_t *)nullptr,
                       pl->Htot, sc->hot_x.as<float>());
}

template <int ABL, int D = PB_ACC_DEPTH, bool BF = false, bool H16 = false>
void pb_launch_accum(const PbPlan *pl, PbScratch *sc, const PbItem *items, uint32_t count, float *x_out, float *scores,
                     const uint32_t *outdeg, float base, float damping, hipStream_t st, bool any_order = false)
{
    (void)pb_launch_flags(pb_accum_kernel<ABL, D, BF, H16>, dim3(count), dim3(PB_ACC_BLOCK),
                          (size_t)pl->Racc * 8 + 16 + (((size_t)pl->H + 3) & ~(size_t)3) * 4 * (pl->T > 1 ? 2 : 1), st, any_order,
                          sc->vals, pl->p2_dst.as<uint16_t>(), items, pl->hot_ent.as<uint32_t>(), pl->hbin_v.as<uint32_t>(),
                          sc->hot_x.as<float>(), pl->H, pl->T, pl->Htot, sc->partials.as<unsigned long long>(),
                          sc->tickets.as<uint32_t>(), pl->cidx.as<uint16_t>(), outdeg, scores, x_out, sc->bin_err.as<double>(),
                          pl->n_local, pl->R, pl->Racc, base, damping, pl->hot_base.as<uint16_t>(), sc->fold);
}

// GM_PB_ABLATE = 10*accumulate variant + bin variant; 0 = the product kernels (re-read per call so that
// tools/ablate.py can switch variants on one resident graph)
// returns whether the launch gathered the hot sources' out_scores as well (whole sweeps of the product kernel)
static bool pb_bin_dispatch(const PbPlan *pl, PbScratch *sc, const float *x_in, uint32_t w_first, uint32_t w_count,
                            hipStream_t st, const uint32_t *item_list, bool fold_hot, hipEvent_t stop)
{
    if (w_count == 0)
        return false;
    fold_hot = fold_hot && pl->Htot && (uint64_t)w_count * PB_BIN_BLOCK >= pl->Htot;
#ifdef GM_MEASURE // the measurement library only (make measure; tools/ablate.py): variants that leave one ingredient out
    const int abl = item_list ? 0 : pb_env("GM_PB_ABLATE", 0) % 10;
    fold_hot = fold_hot && abl == 0;
    if (pl->s_log == 15 && (abl == 3 || abl == 5)) {
        abl == 3 ? pb_launch_bin<3, 15>(pl, sc, x_in, w_first, w_count, st) : pb_launch_bin<5, 15>(pl, sc, x_in, w_first, w_count, st);
        return false;
    }
    if (pl->s_log == 14 && abl) {
        switch (abl) {
        case 1: pb_launch_bin<1, 14>(pl, sc, x_in, w_first, w_count, st); return false;
        case 3: pb_launch_bin<3, 14>(pl, sc, x_in, w_first, w_count, st); return false;
        case 4: pb_launch_bin<4, 14>(pl, sc, x_in, w_first, w_count, st); return false;
        case 5: pb_launch_bin<5, 14>(pl, sc, x_in, w_first, w_count, st); return false;
        default: break;
        }
    }
#endif
    if (pl->s_log == 15)
        pb_launch_bin<0, 15>(pl, sc, x_in, w_first, w_count, st, item_list, fold_hot, stop);
    else
        pb_launch_bin<0, 14>(pl, sc, x_in, w_first, w_count, st, item_list, fold_hot, stop);
    return fold_hot;
}

static void pb_accum_dispatch(const PbPlan *pl, PbScratch *sc, const PbItem *items, uint32_t count, float *x_out,
                              float *scores, const uint32_t *outdeg, float base, float damping, hipStream_t st,
                              bool any_order = false)
{
    if (count == 0)
        return;
#ifdef GM_MEASURE // the measurement library only: a phase left out (GM_PB_ABLATE / 10), other ring depths, branch-free padding
    if (pl->hot16) {
        switch (pb_env("GM_PB_ABLATE", 0) / 10) {
        case 5: pb_launch_accum<5, PB_ACC_DEPTH, false, true>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        case 6: pb_launch_accum<6, PB_ACC_DEPTH, false, true>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        default: break;
        }
    } else {
        switch (pb_env("GM_PB_ABLATE", 0) / 10) {
        case 3: pb_launch_accum<3>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        case 4: pb_launch_accum<4>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        case 5: pb_launch_accum<5>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        case 6: pb_launch_accum<6>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        default: break;
        }
        switch (pb_env("GM_PB_ACC_DEPTH", PB_ACC_DEPTH)) { // register groups of the value stream in flight
        case 2: pb_launch_accum<0, 2>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        case 6: pb_launch_accum<0, 6>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st); return;
        default: break;
        }
        if (pb_env("GM_PB_ACC_BRANCHFREE", 0)) {
            pb_launch_accum<0, PB_ACC_DEPTH, true>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st);
            return;
        }
    }
#endif
    if (pl->hot16) // a plan with 2-byte hot records
        pb_launch_accum<0, PB_ACC_DEPTH, false, true>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st, any_order);
    else
        pb_launch_accum<0>(pl, sc, items, count, x_out, scores, outdeg, base, damping, st, any_order);
}

// workgroups of pb_hubseq_kernel (GM_PB_SEQ_WGS: fewer than one per group, each looping over several)
static uint32_t pb_seq_wgs(const PbPlan *pl)
{
    const uint32_t n_seq = pl->G - pl->G_long;
    const uint32_t want = (uint32_t)pb_env_m("GM_PB_SEQ_WGS", 0);
    return want && want < n_seq ? want : n_seq;
}

// workgroups the hub launches of a sweep have in all (pb_hub_dispatch)
[[maybe_unused]] static uint32_t pb_hub_workgroups(const PbPlan *pl)
{
    if (!pl->G)
        return 0;
    if (pl->hub_csr)
        return pl->n_long_items;
    return (pl->G_long ? pl->n_long_items : 0u) + (pl->G > pl->G_long ? pb_seq_wgs(pl) : 0u);
}

// every hub group of the plan (its value-stream part must have been written: after the bin kernel)
// `inline_any`: both kernels on `st`, the second one (and, by the caller, the accumulate kernel behind them) launched in any
// order; returns whether anything was launched (the first launch behind the bin kernel must be an ordered one)
struct PbHubSubset { // the hub work of one part (hub_by_part); null lists with a count = nothing of that kind
    const uint32_t *seq_list = nullptr;
    uint32_t n_seq = 0;
    const uint32_t *long_list = nullptr;
    uint32_t n_long = 0;
    unsigned long long *ticket = nullptr;
};

static bool pb_hub_dispatch(const PbPlan *pl, PbScratch *sc, float *x_out, float *scores, const uint32_t *outdeg, float base,
                            float damping, hipStream_t st, bool inline_any = false, const PbHubSubset *sub = nullptr)
{
    double *gerr = sc->bin_err.as<double>() + pl->B;
    const PbHubItem *items = pl->hub_items.as<PbHubItem>();
#ifdef GM_MEASURE
    const int skip = pb_env("GM_PB_HUB_SKIP", 0); // measurement (wrong results by design): 1 = no pb_hubseq_kernel, 2 = no long rows
#else
    constexpr int skip = 0;
#endif
    // GM_PB_LONG_WGS / GM_PB_SEQ_WGS: workgroups of the two kernels (0 = one per row / group)
    const uint32_t n_seq = sub ? sub->n_seq : pl->G - pl->G_long;
    const uint32_t seq_wgs = sub ? sub->n_seq : pb_seq_wgs(pl);
    const uint32_t n_long_launch = sub ? sub->n_long : pl->n_long_items;
    const uint32_t *seq_list = sub ? sub->seq_list : nullptr, *long_list = sub ? sub->long_list : nullptr;
    // the long rows: one workgroup per item (a row, or a few passes of a longer one), exactly n_long_items draws of the counter
    unsigned long long *l_ticket = sc->long_state.as<unsigned long long>(), *l_handoff = l_ticket + 1;
    if (sub)
        l_ticket = sub->ticket; // (a counter per part: each is drawn exactly as often as its part has items, launch after launch)
    float *l_sbs = reinterpret_cast<float *>(l_handoff + pl->n_long_items);
    const PbLongItem *l_items = pl->long_items.as<PbLongItem>();
    const uint32_t l_epoch = (pl->hub_csr ? pl->n_hub : pl->G_long) && !(skip & 2) ? ++sc->long_epoch : 0u;
    if (pl->hub_csr) { // lists that are not ascending: every hub row in CSR order through its index
        hipLaunchKernelGGL(pb_hublong_kernel<true>, dim3(pl->n_long_items), dim3(PB_LONG_WG), 0, st, sc->vals, pl->p2_dst.as<uint16_t>(),
                           pl->hub_gidx.as<uint32_t>(), pl->csr_items.as<PbHubItem>(), l_items, pl->n_long_items, l_ticket, l_handoff,
                           l_sbs, l_epoch, pl->hub_rows.as<uint32_t>(), outdeg, scores, x_out, gerr, base, damping, sc->fold,
                           (const uint32_t *)nullptr);
        return true;
    }
    const uint32_t v_safe = (uint32_t)(pl->Mv >= 4 ? (pl->Mv - 4) & ~3ull : 0), h_safe = (uint32_t)(pl->Mhh ? pl->Mhh - 1u : 0u);
#ifdef GM_MEASURE // launches without the barrier bit on one stream (gfx950 serialises them: DESIGN.md)
    if (inline_any) {
        bool launched = false;
        if (pl->G_long && !(skip & 2)) {
            (void)pb_launch_flags(pb_hublong_kernel<false>, dim3(pl->n_long_items), dim3(PB_LONG_WG), 0, st, launched, sc->vals,
                                  pl->p2_dst.as<uint16_t>(), (const uint32_t *)nullptr, pl->long_rows.as<PbHubItem>(), l_items, pl->n_long_items, l_ticket, l_handoff, l_sbs, l_epoch,
                                  pl->hub_rows.as<uint32_t>(), outdeg, scores, x_out, gerr, base, damping, PbErrFold{}, (const uint32_t *)nullptr);
            launched = true;
        }
        if (pl->G > pl->G_long && !(skip & 1)) {
            (void)pb_launch_flags(pb_hubseq_kernel<PB_SEQ_WG>, dim3(seq_wgs), dim3(PB_SEQ_WG), 0, st, launched, sc->vals,
                                  pl->p2_dst.as<uint16_t>(), items + pl->G_long, pl->seq_blk_first.as<uint32_t>(),
                                  pl->seq_blk.as<uint4>(), pl->seq_rows.as<uint32_t>(), pl->hh_ent.as<uint32_t>(),
                                  sc->hot_x.as<float>(), pl->hub_rows.as<uint32_t>(), outdeg, scores, x_out, gerr, base, damping,
                                  v_safe, h_safe, n_seq, (uint32_t)pb_env_m("GM_PB_SEQ_PRIO", 1), PbErrFold{}, (const uint32_t *)nullptr);
            launched = true;
        }
        return launched;
    }
#endif
    // the long rows on a stream of their own beside the other groups (when there are both)
    const bool have_long = pl->G_long && n_long_launch && !(skip & 2), have_seq = pl->G > pl->G_long && n_seq && !(skip & 1);
    const bool own = have_long && have_seq && sc->chain;
    hipStream_t ls = own ? sc->chain : st;
    if (have_long) {
        if (own) {
            (void)hipEventRecord(sc->ev_chain_fork, st);
            (void)hipStreamWaitEvent(ls, sc->ev_chain_fork, 0);
        }
        hipLaunchKernelGGL(pb_hublong_kernel<false>, dim3(n_long_launch), dim3(PB_LONG_WG), 0, ls, sc->vals, pl->p2_dst.as<uint16_t>(),
                           (const uint32_t *)nullptr, pl->long_rows.as<PbHubItem>(), l_items, n_long_launch, l_ticket, l_handoff, l_sbs, l_epoch, pl->hub_rows.as<uint32_t>(), outdeg, scores,
                           x_out, gerr, base, damping, sc->fold, long_list);
        if (own)
            (void)hipEventRecord(sc->ev_chain_join, ls);
    }
    if (have_seq) {
        // (GM_PB_SEQ_WIDE=0 / 1, measurements: never / always the 512-thread workgroups)
        const bool wide = pb_env_m("GM_PB_SEQ_WIDE", (sub != nullptr || pl->x_len != pl->n_local) ? 1 : 0) != 0;
        auto *kernel = wide ? pb_hubseq_kernel<PB_SEQ_WG_WIDE> : pb_hubseq_kernel<PB_SEQ_WG>;
        hipLaunchKernelGGL(kernel, dim3(seq_wgs), dim3(wide ? PB_SEQ_WG_WIDE : PB_SEQ_WG), 0, st, sc->vals, pl->p2_dst.as<uint16_t>(),
                           items + pl->G_long, pl->seq_blk_first.as<uint32_t>(), pl->seq_blk.as<uint4>(), pl->seq_rows.as<uint32_t>(),
                           pl->hh_ent.as<uint32_t>(), sc->hot_x.as<float>(), pl->hub_rows.as<uint32_t>(), outdeg, scores, x_out, gerr,
                           base, damping, v_safe, h_safe, n_seq, (uint32_t)pb_env_m("GM_PB_SEQ_PRIO", 1), sc->fold, seq_list);
    }
    if (own)
        (void)hipStreamWaitEvent(st, sc->ev_chain_join, 0);
    return have_long || have_seq;
}

static void pb_hot_dispatch(const PbPlan *pl, PbScratch *sc, const float *x_in, hipStream_t st)
{
    if (pl->Htot)
        hipLaunchKernelGGL(pb_hot_gather_kernel, dim3(div_up(pl->Htot, 256)), dim3(256), 0, st, x_in,
                           pl->hot_ids.as<uint32_t>(), pl->Htot, sc->hot_x.as<float>());
}

// GM_PB_VALS_OFFSET=<KiB> (measurements only; needs GM_PB_VALS_SLACK at creation): where inside its allocation the value
// stream starts, re-read at every launch
static void pb_apply_vals_offset(const PbPlan *pl, PbScratch *sc)
{
#ifndef GM_MEASURE
    (void)pl, (void)sc;
#else
    if (const char *off = getenv("GM_PB_VALS_OFFSET")) {
        const size_t bytes = (size_t)atoll(off) << 10;
        if (bytes + (size_t)pl->Mv * 4 <= sc->vals_raw.bytes)
            sc->vals = reinterpret_cast<float *>(sc->vals_raw.as<char>() + bytes);
    }
#endif
}

// err_out != null: the sweep's error as well (pb_err_fold: no launch of its own); *folded_out says whether that happened —
// if not (nothing to launch, the measurement library) the caller runs pb_sweep_error
int pb_sweep_main(const PbPlan *pl, PbScratch *sc, const float *x_in, float *x_out, float *scores,
                  const uint32_t *outdeg, float base, float damping, hipStream_t st, double *err_out, bool *folded_out)
{
    if (folded_out)
        *folded_out = false;
    struct FoldScope { // the launches below, and only they, carry the fold
        PbScratch *sc;
        ~FoldScope() { sc->fold = PbErrFold{}; }
    } fold_scope{sc};
#ifndef GM_MEASURE
    if (err_out && folded_out && pl->NI && sc->fold_ctr.p && pb_env_m("GM_PB_FOLD_ERR", 1)) {
        sc->fold.ctr = sc->fold_ctr.as<uint32_t>();
        sc->fold.total = pl->NI + pb_hub_workgroups(pl);
        sc->fold.count = pl->B + pl->err_slots;
        sc->fold.slots = sc->bin_err.as<double>();
        sc->fold.out = err_out;
        *folded_out = true;
    }
#endif
    pb_apply_vals_offset(pl, sc);
    // the hub groups need little LDS: on a second stream their workgroups run beside those of the ordinary bins
    // (measured at scale 26 / 22 on one box: 3.12 / 3.24 ms forked vs 3.33 / 3.44 ms in line, 0.216 vs 0.252 ms;
    // with the 85-VGPR version of the kernel the two could not share a CU and forking gained nothing)
    const bool fork = pl->G && sc->side && pl->hub_edges >= (1u << 20) && pb_env("GM_PB_HUB_FORK", 1);
    // GM_PB_FORK_STOP=1 (round 6, measured: see DESIGN.md): the fork event is the bin kernel's own completion instead of a
    // marker recorded behind it — when the bin launch gathers the hot sources as well (nothing else stands between the two)
    const bool want_fold_hot = pb_env_m("GM_PB_FOLD_HOT", 1) != 0 && pl->Htot && (uint64_t)pl->NW * PB_BIN_BLOCK >= pl->Htot;
    bool fork_by_stop = fork && pl->NW && want_fold_hot && pb_env_m("GM_PB_FORK_STOP", 0) != 0;
#ifdef GM_MEASURE
    fork_by_stop = false;
#endif
    if (!pb_bin_dispatch(pl, sc, x_in, 0, pl->NW, st, nullptr, pb_env_m("GM_PB_FOLD_HOT", 1) != 0, fork_by_stop ? sc->ev_fork : nullptr))
        pb_hot_dispatch(pl, sc, x_in, st); // (in front of the accumulate kernel, which reads hot_x; the bin kernel does not)
#ifdef GM_MEASURE
    if (pl->G && pb_env("GM_PB_ANYORDER", 0)) {
        const bool any = pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, st, true);
        pb_accum_dispatch(pl, sc, pl->items.as<PbItem>(), pl->NI, x_out, scores, outdeg, base, damping, st, any);
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
#endif
    if (fork) {
        if (!fork_by_stop)
            GM_HIP(hipEventRecord(sc->ev_fork, st));
        GM_HIP(hipStreamWaitEvent(sc->side, sc->ev_fork, 0));
        pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, sc->side);
        GM_HIP(hipEventRecord(sc->ev_join, sc->side));
    } else {
        pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, st);
    }
    pb_accum_dispatch(pl, sc, pl->items.as<PbItem>(), pl->NI, x_out, scores, outdeg, base, damping, st);
    if (fork)
        GM_HIP(hipStreamWaitEvent(st, sc->ev_join, 0));
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// ---- partitioned sweeps in pieces (overlap of the exchange with the work) ----------
uint32_t pb_rows_per_bin(const PbPlan *pl) { return pl->R; }
uint32_t pb_source_tile(const PbPlan *pl) { return 1u << pl->s_log; }

// bins [row_splits[k] / R, row_splits[k+1] / R) form part k; row_splits[0] = 0, the last one = n_local,
// the inner ones multiples of R
int pb_set_parts(const PbPlan *pl, PbScratch *sc, const uint64_t *row_splits, uint32_t n_parts, bool hub_by_part)
{
    GM_CHECK(n_parts >= 1 && row_splits && row_splits[0] == 0 && row_splits[n_parts] == pl->n_local, GM_ERR_INVALID,
             "gm_pr_set_parts: the splits must start at 0 and end at the %u local rows", pl->n_local);
    std::vector<uint32_t> first_bin(n_parts + 1);
    for (uint32_t k = 0; k <= n_parts; ++k) {
        GM_CHECK(k == 0 || row_splits[k] >= row_splits[k - 1], GM_ERR_INVALID, "gm_pr_set_parts: splits must ascend");
        const bool at_end = row_splits[k] == pl->n_local; // trailing groups may be empty
        GM_CHECK(at_end || (row_splits[k] < pl->n_local && row_splits[k] % pl->R == 0), GM_ERR_INVALID,
                 "gm_pr_set_parts: split %llu is not a multiple of the %u rows of a bin",
                 (unsigned long long)row_splits[k], pl->R);
        first_bin[k] = at_end ? pl->B : (uint32_t)(row_splits[k] / pl->R);
    }
    std::vector<PbItem> ordered;
    ordered.reserve(pl->items_host.size());
    sc->part_off.assign(n_parts + 1, 0);
    for (uint32_t k = 0; k < n_parts; ++k) {
        sc->part_off[k] = (uint32_t)ordered.size();
        for (const PbItem &it : pl->items_host) // keeps the longest-first order inside the part
            if (it.bin >= first_bin[k] && it.bin < first_bin[k + 1])
                ordered.push_back(it);
    }
    sc->part_off[n_parts] = (uint32_t)ordered.size();
    GM_CHECK(ordered.size() == pl->items_host.size(), GM_ERR_INVALID, "gm_pr_set_parts: the parts do not cover every bin");
    GM_TRY(sc->part_items.alloc(ordered.size() * sizeof(PbItem)));
    GM_HIP(hipMemcpy(sc->part_items.p, ordered.data(), ordered.size() * sizeof(PbItem), hipMemcpyHostToDevice));
    // hub rows with the part their rows lie in (block-Gauss-Seidel sweeps: a hub row sees this sweep's values of the blocks
    // before it, like every other row).  A lane-walk group goes with the part of its FIRST row (its other rows lie in that
    // part or later ones: finished no later than their own part's turn).  Lists that are not ascending (hub_csr) stay with part 0.
    sc->hub_by_part = false;
    if (hub_by_part && pl->G && !pl->hub_csr && n_parts > 1) {
        std::vector<uint32_t> hub_rows_host(pl->n_hub);
        GM_HIP(hipMemcpy(hub_rows_host.data(), pl->hub_rows.p, (size_t)pl->n_hub * 4, hipMemcpyDeviceToHost));
        auto part_of = [&](uint32_t row) {
            uint32_t k = 0;
            while (k + 1 < n_parts && row_splits[k + 1] <= row)
                ++k;
            return k;
        };
        std::vector<std::vector<uint32_t>> seq(n_parts), lng(n_parts);
        for (uint32_t gi = 0; gi + pl->G_long < pl->G; ++gi)
            seq[part_of(hub_rows_host[pl->hub_items_host[pl->G_long + gi].row0])].push_back(gi);
        for (uint32_t i = 0; i < pl->long_items_host.size(); ++i) // (row by row, every row's items in pass order)
            lng[part_of(hub_rows_host[pl->long_rows_host[pl->long_items_host[i].row].row0])].push_back(i);
        std::vector<uint32_t> flat_s, flat_l;
        sc->part_seq_off.assign(n_parts + 1, 0);
        sc->part_long_off.assign(n_parts + 1, 0);
        for (uint32_t k = 0; k < n_parts; ++k) {
            sc->part_seq_off[k] = (uint32_t)flat_s.size();
            sc->part_long_off[k] = (uint32_t)flat_l.size();
            flat_s.insert(flat_s.end(), seq[k].begin(), seq[k].end());
            flat_l.insert(flat_l.end(), lng[k].begin(), lng[k].end());
        }
        sc->part_seq_off[n_parts] = (uint32_t)flat_s.size();
        sc->part_long_off[n_parts] = (uint32_t)flat_l.size();
        GM_TRY(sc->part_seq_list.alloc((flat_s.size() ? flat_s.size() : 1) * 4));
        GM_TRY(sc->part_long_list.alloc((flat_l.size() ? flat_l.size() : 1) * 4));
        GM_TRY(sc->part_long_tickets.alloc((size_t)n_parts * 8));
        if (!flat_s.empty())
            GM_HIP(hipMemcpy(sc->part_seq_list.p, flat_s.data(), flat_s.size() * 4, hipMemcpyHostToDevice));
        if (!flat_l.empty())
            GM_HIP(hipMemcpy(sc->part_long_list.p, flat_l.data(), flat_l.size() * 4, hipMemcpyHostToDevice));
        GM_HIP(hipMemset(sc->part_long_tickets.p, 0, (size_t)n_parts * 8));
        GM_HIP(hipDeviceSynchronize());
        sc->hub_by_part = true;
    }
    return GM_OK;
}

// propagates x[x_lo, x_hi) (whole source tiles; x_hi may also be the end of the vector) into the value stream
int pb_sweep_bin_range(const PbPlan *pl, PbScratch *sc, const float *x_in, uint64_t x_lo, uint64_t x_hi, hipStream_t st)
{
    const uint64_t S = 1ull << pl->s_log;
    if (x_hi > pl->x_len)
        x_hi = pl->x_len;
    GM_CHECK(x_lo % S == 0 && (x_hi % S == 0 || x_hi == pl->x_len), GM_ERR_INVALID,
             "gm_pr_sweep_bin: [%llu, %llu) is not a range of whole source tiles (%llu)", (unsigned long long)x_lo,
             (unsigned long long)x_hi, (unsigned long long)S);
    const uint64_t tile_lo = x_lo >> pl->s_log, tile_hi = (x_hi + S - 1) >> pl->s_log;
    if (pl->NW == 0 || tile_lo >= tile_hi)
        return GM_OK;
    const uint32_t w0 = pl->wg_first_host[tile_lo], w1 = pl->wg_first_host[tile_hi > pl->NT ? pl->NT : tile_hi];
    pb_apply_vals_offset(pl, sc);
    pb_bin_dispatch(pl, sc, x_in, w0, w1 - w0, st);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// Regions of the x vector as LISTS of tile ranges: range i = [x_lo[i], x_hi[i]) (whole source tiles; x_hi may be the end
// of the vector) belongs to region reg[i].  A partitioned run keeps x rank-major — ascending node ids, which the hub
// rows' summation order follows — and exchanges row group k of every rank as region k: P ranges, one launch.
int pb_set_regions(const PbPlan *pl, PbScratch *sc, const uint64_t *x_lo, const uint64_t *x_hi, const uint32_t *reg,
                   uint32_t count, uint32_t n_regions)
{
    const uint64_t S = 1ull << pl->s_log;
    std::vector<std::vector<uint32_t>> items(n_regions);
    for (uint32_t i = 0; i < count; ++i) {
        uint64_t lo = x_lo[i], hi = x_hi[i] > pl->x_len ? pl->x_len : x_hi[i];
        GM_CHECK(reg[i] < n_regions, GM_ERR_INVALID, "gm_pr_set_bin_regions: range %u names region %u of %u", i, reg[i], n_regions);
        GM_CHECK(lo % S == 0 && (hi % S == 0 || hi == pl->x_len) && lo <= hi, GM_ERR_INVALID,
                 "gm_pr_set_bin_regions: [%llu, %llu) is not a range of whole source tiles (%llu)", (unsigned long long)lo,
                 (unsigned long long)hi, (unsigned long long)S);
        if (pl->NW == 0 || lo >= hi)
            continue;
        const uint64_t t_lo = lo >> pl->s_log, t_hi = (hi + S - 1) >> pl->s_log;
        for (uint32_t w = pl->wg_first_host[t_lo]; w < pl->wg_first_host[t_hi > pl->NT ? pl->NT : t_hi]; ++w)
            items[reg[i]].push_back(w);
    }
    std::vector<uint32_t> flat;
    sc->region_off.assign(n_regions + 1, 0);
    for (uint32_t r = 0; r < n_regions; ++r) {
        sc->region_off[r] = (uint32_t)flat.size();
        flat.insert(flat.end(), items[r].begin(), items[r].end());
    }
    sc->region_off[n_regions] = (uint32_t)flat.size();
    GM_TRY(sc->region_items.alloc((flat.size() ? flat.size() : 1) * 4));
    if (!flat.empty())
        GM_HIP(hipMemcpy(sc->region_items.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    return GM_OK;
}

// propagates the tiles of region `region` (pb_set_regions) into the value stream: one launch
int pb_sweep_bin_region(const PbPlan *pl, PbScratch *sc, const float *x_in, uint32_t region, hipStream_t st)
{
    GM_CHECK(region + 1 < sc->region_off.size(), GM_ERR_INVALID, "gm_pr_sweep_bin_region: region %u of %zu (call gm_pr_set_bin_regions first)",
             region, sc->region_off.empty() ? (size_t)0 : sc->region_off.size() - 1);
    const uint32_t i0 = sc->region_off[region], i1 = sc->region_off[region + 1];
    pb_apply_vals_offset(pl, sc);
    pb_bin_dispatch(pl, sc, x_in, 0, i1 - i0, st, sc->region_items.as<uint32_t>() + i0);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// stages the hot sources' values of x_in for the accumulate launches of this sweep (needs the whole vector)
int pb_sweep_hot(const PbPlan *pl, PbScratch *sc, const float *x_in, hipStream_t st)
{
    pb_hot_dispatch(pl, sc, x_in, st);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

// accumulates and finishes the rows of part `part`, after every tile of x has been propagated and the hot
// sources staged (stage_hot != 0: stage them here first — the simple in-order schedule does that on part 0)
int pb_sweep_accum_part(const PbPlan *pl, PbScratch *sc, const float *x_in, float *x_out, float *scores,
                        const uint32_t *outdeg, float base, float damping, uint32_t part, int stage_hot, hipStream_t st)
{
    GM_CHECK(sc->part_off.size() >= 2 && part + 1 < sc->part_off.size(), GM_ERR_INVALID,
             "gm_pr_sweep_accum: part %u of %zu (call gm_pr_set_parts first)", part,
             sc->part_off.empty() ? (size_t)0 : sc->part_off.size() - 1);
    if (stage_hot)
        pb_hot_dispatch(pl, sc, x_in, st);
    // The hub rows (any of them may lie in any part's rows) are summed by launches that go out with part 0 — BESIDE its
    // accumulate kernel, on the engine's side streams, as in a whole sweep (round 5: until then they ran in line in front of
    // it, and the rank's first accumulate piece waited 0.22 ms for the lane walks at scale 26 / 8 ranks) — and EVERY part's
    // stream waits for them behind its own accumulate kernel: whatever the caller enqueues next on that stream (the
    // exchange of the part's rows) sees its hub rows finished.
    const bool fork = pl->G && sc->side && pl->hub_edges >= (1u << 20) && pb_env("GM_PB_HUB_FORK", 1);
    bool any = false;
    if (sc->hub_by_part && part + 1 < sc->part_seq_off.size()) {
        // this part's hub rows, beside this part's accumulate kernel (block-Gauss-Seidel sweeps: the parts follow each other
        // on one stream, so the events are free again when the next part records them)
        PbHubSubset sub;
        sub.seq_list = sc->part_seq_list.as<uint32_t>() + sc->part_seq_off[part];
        sub.n_seq = sc->part_seq_off[part + 1] - sc->part_seq_off[part];
        sub.long_list = sc->part_long_list.as<uint32_t>() + sc->part_long_off[part];
        sub.n_long = sc->part_long_off[part + 1] - sc->part_long_off[part];
        sub.ticket = sc->part_long_tickets.as<unsigned long long>() + part;
        const bool work = sub.n_seq || sub.n_long;
        if (work && fork) {
            GM_HIP(hipEventRecord(sc->ev_fork, st));
            GM_HIP(hipStreamWaitEvent(sc->side, sc->ev_fork, 0));
            pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, sc->side, false, &sub);
            GM_HIP(hipEventRecord(sc->ev_join, sc->side));
        } else if (work) {
            pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, st, false, &sub);
        }
        const uint32_t i0 = sc->part_off[part], i1 = sc->part_off[part + 1];
        pb_accum_dispatch(pl, sc, sc->part_items.as<PbItem>() + i0, i1 - i0, x_out, scores, outdeg, base, damping, st);
        if (work && fork)
            GM_HIP(hipStreamWaitEvent(st, sc->ev_join, 0));
        GM_HIP(hipGetLastError());
        return GM_OK;
    }
    if (part == 0) {
#ifdef GM_MEASURE
        if (pl->G && pb_env("GM_PB_ANYORDER", 0)) {
            any = pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, st, true);
        } else
#endif
        if (fork) {
            GM_HIP(hipEventRecord(sc->ev_fork, st));
            GM_HIP(hipStreamWaitEvent(sc->side, sc->ev_fork, 0));
            pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, sc->side);
            GM_HIP(hipEventRecord(sc->ev_join, sc->side));
        } else {
            pb_hub_dispatch(pl, sc, x_out, scores, outdeg, base, damping, st); // (few hub edges: in line, no events)
            if (pl->G && sc->ev_join)
                GM_HIP(hipEventRecord(sc->ev_join, st));
        }
    }
    const uint32_t i0 = sc->part_off[part], i1 = sc->part_off[part + 1];
    pb_accum_dispatch(pl, sc, sc->part_items.as<PbItem>() + i0, i1 - i0, x_out, scores, outdeg, base, damping, st, any);
    if (pl->G && sc->ev_join && !any && (fork || part != 0)) // (recorded by this sweep's part 0: the parts are enqueued in order)
        GM_HIP(hipStreamWaitEvent(st, sc->ev_join, 0));
    GM_HIP(hipGetLastError());
    return GM_OK;
}

int pb_sweep_error(const PbPlan *pl, PbScratch *sc, double *err_out, hipStream_t st)
{
    hipLaunchKernelGGL(pb_err_kernel, dim3(1), dim3(1024), 0, st, sc->bin_err.as<double>(), pl->B + pl->err_slots, err_out);
    GM_HIP(hipGetLastError());
    return GM_OK;
}

void warm_pagerank_pb() // (common.hpp: the code object of this file, loaded ahead of the first plan build)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&pb_hot_gather_kernel)) != hipSuccess)
        (void)hipGetLastError();
}

} // namespace gm
