// pagerank.hpp — shared between the two PageRank sweep engines (pagerank.hip: pull tiles,
// pagerank_pb.hip: propagation blocking).
#pragma once
#include "common.hpp"
#include "device_utils.hpp"

namespace gm {

// ---- per-node arithmetic, exactly the reference's f32 ops (no FMA contraction) ----------------
// crates/algos/src/page_rank.rs:149-159
__device__ __forceinline__ float pr_new_score(float base, float damping, float incoming)
{
    return __fadd_rn(base, __fmul_rn(damping, incoming));
}

__device__ __forceinline__ double pr_finalize(uint32_t r, float incoming, float base, float damping,
                                              const uint32_t *__restrict__ outdeg, float *__restrict__ scores,
                                              float *__restrict__ x_out)
{
    const float old = scores[r];
    const float nw = pr_new_score(base, damping, incoming);
    scores[r] = nw;
    x_out[r] = __fdiv_rn(nw, (float)outdeg[r]); // out_degree 0 -> +inf, never gathered (page_rank.rs:78,158)
    return fabs((double)__fsub_rn(nw, old));
}

struct PbPlan;    // pagerank_pb.hip: immutable layout, cached in the gm_csr handle
struct PbScratch; // per-engine mutable buffers (value stream, partial sums, tickets, errors, hot values)
int pb_plan_get(const gm_csr *csr, uint64_t x_len, std::shared_ptr<const PbPlan> *out); // build on first use, then shared
// `early`: a buffer allocated before the plan was built (may be null or too small): it becomes the value stream
int pb_scratch_create(const PbPlan *plan, PbScratch **out, DevBuf *early = nullptr);
void pb_scratch_destroy(PbScratch *scratch);
int pb_sweep_main(const PbPlan *plan, PbScratch *scratch, const float *x_in, float *x_out, float *scores,
                  const uint32_t *outdeg, float base, float damping, hipStream_t st, double *err_out = nullptr,
                  bool *folded_out = nullptr);
int pb_sweep_error(const PbPlan *plan, PbScratch *scratch, double *err_out, hipStream_t st);
// a sweep in pieces, for partitioned runs that overlap the exchange of x with the work (pagerank_pb.hip)
uint32_t pb_rows_per_bin(const PbPlan *plan);
uint32_t pb_source_tile(const PbPlan *plan);
int pb_set_parts(const PbPlan *plan, PbScratch *scratch, const uint64_t *row_splits, uint32_t n_parts, bool hub_by_part = false);
int pb_sweep_bin_range(const PbPlan *plan, PbScratch *scratch, const float *x_in, uint64_t x_lo, uint64_t x_hi,
                       hipStream_t st);
int pb_set_regions(const PbPlan *plan, PbScratch *scratch, const uint64_t *x_lo, const uint64_t *x_hi, const uint32_t *reg,
                   uint32_t count, uint32_t n_regions);
int pb_sweep_bin_region(const PbPlan *plan, PbScratch *scratch, const float *x_in, uint32_t region, hipStream_t st);
int pb_sweep_hot(const PbPlan *plan, PbScratch *scratch, const float *x_in, hipStream_t st);
int pb_sweep_accum_part(const PbPlan *plan, PbScratch *scratch, const float *x_in, float *x_out, float *scores,
                        const uint32_t *outdeg, float base, float damping, uint32_t part, int stage_hot, hipStream_t st);
uint64_t pb_work_items(const PbPlan *plan);
void pb_plan_info(const PbPlan *plan, const PbScratch *scratch, uint64_t *info, uint32_t count);

} // namespace gm


struct gm_pr {
    const gm_csr *csr = nullptr;
    uint64_t n_global = 0, row_begin = 0, x_len = 0;
    uint32_t n_local = 0, m = 0, T = 0, G = 0;
    int engine = GM_PR_ENGINE_PULL;
    const uint32_t *outdeg = nullptr;
    float damping = 0.85f, base = 0.0f, init = 0.0f;
    gm::DevBuf tile_row, head, tail, tile_err, blk_err, ticket; // pull engine
    std::shared_ptr<const gm::PbPlan> pb_keep;                  // propagation-blocking engine: shared plan (cached in the csr)
    const gm::PbPlan *pb = nullptr;
    gm::PbScratch *pb_scratch = nullptr;
    ~gm_pr()
    {
        if (pb_scratch)
            gm::pb_scratch_destroy(pb_scratch);
    }
};
