/*
 * graph_oracle.c — CPU restatement of the neo4j-labs/graph `crates/algos` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the timed CPU baseline.  The product path (graph_amd/) never
 * links, imports or calls it.
 *
 * Parity status: PINNED for PageRank (sequential mode), triangle count, delta-stepping
 * and CSR construction by the reference's own golden vectors (tests/test_oracle_golden.py);
 * WCC component ids and multi-threaded PageRank are pinned by no reference test
 * ("parity unpinned" at that granularity) — WCC is nevertheless fully determined
 * (label = minimum vertex id of the weakly connected component).
 *
 * The reference is Rust and cannot be compiled in this environment (no rustc/cargo), so
 * each function restates the reference algorithm and cites the file:line it follows
 * (paths relative to the reference repository root).
 *
 * Node ids are u32 (the reference's `--use-32-bit` / NI = u32 instantiation); offsets are
 * u32 as in the reference's Csr<NI, NI, EV> (crates/builder/src/graph/csr.rs:58-61).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -pthread -shared).
 * -ffp-contract=off matters: the reference computes base + d*s with separate f32 mul
 * and add; an FMA changes the last bit of the README golden vector.
 */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <float.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Synthetic input: R-MAT / Graph500 Kronecker generator (no reference counterpart; the
 * reference downloads LDBC files, crates/builder/benches/common/mod.rs:15-41).  Pure integer
 * arithmetic so the GPU generator (graph_amd/csrc/rmat.hip) reproduces it bit-for-bit.
 * A=0.57 B=0.19 C=0.19 D=0.05, one counter-based splitmix64 draw per two levels, vertex ids
 * scrambled by a bijection on `scale` bits.
 * ------------------------------------------------------------------------------------------ */
static inline uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define RMAT_T_A 2448131358u   /* floor(0.57 * 2^32) */
#define RMAT_T_AB 3264175145u  /* floor(0.76 * 2^32) */
#define RMAT_T_ABC 4080218931u /* floor(0.95 * 2^32) */

static inline uint32_t orc_scramble(uint32_t x, uint32_t scale, uint64_t seed)
{
    /* bijection on [0, 2^scale): odd multiply, xor-shift, odd multiply, add */
    const uint32_t mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    const uint32_t k1 = (uint32_t)(orc_splitmix64(seed) | 1u);
    const uint32_t k2 = (uint32_t)(orc_splitmix64(seed + 1) | 1u);
    const uint32_t k3 = (uint32_t)orc_splitmix64(seed + 2);
    const uint32_t sh = (scale + 1) / 2;
    x = (x * k1) & mask;
    x ^= x >> sh;
    x = (x * k2) & mask;
    x ^= x >> sh;
    x = (x + k3) & mask;
    return x;
}

static inline void orc_rmat_edge(uint32_t scale, uint64_t seed, uint64_t idx, uint32_t *s, uint32_t *t)
{
    uint32_t src = 0, dst = 0;
    for (uint32_t lvl = 0; lvl < scale; lvl += 2) {
        uint64_t h = orc_splitmix64(seed ^ (idx * 32u + (lvl >> 1)) * 0xD1342543DE82EF95ull);
        uint32_t r0 = (uint32_t)h, r1 = (uint32_t)(h >> 32);
        uint32_t sb = r0 >= RMAT_T_AB;
        uint32_t db = (r0 >= RMAT_T_A && r0 < RMAT_T_AB) || r0 >= RMAT_T_ABC;
        src = (src << 1) | sb;
        dst = (dst << 1) | db;
        if (lvl + 1 < scale) {
            sb = r1 >= RMAT_T_AB;
            db = (r1 >= RMAT_T_A && r1 < RMAT_T_AB) || r1 >= RMAT_T_ABC;
            src = (src << 1) | sb;
            dst = (dst << 1) | db;
        }
    }
    *s = orc_scramble(src, scale, seed ^ 0x5851F42D4C957F2Dull);
    *t = orc_scramble(dst, scale, seed ^ 0x5851F42D4C957F2Dull);
}

ORC_EXPORT void orc_rmat_edges(uint32_t scale, uint64_t seed, uint64_t first, uint64_t count,
                               uint32_t *src, uint32_t *dst)
{
    for (uint64_t i = 0; i < count; ++i)
        orc_rmat_edge(scale, seed, first + i, &src[i], &dst[i]);
}

/* f32 edge weight in (0,1] for edge index idx: (k+1)/2^24 with k = 24 random bits */
ORC_EXPORT void orc_rmat_weights(uint64_t seed, uint64_t first, uint64_t count, float *w)
{
    for (uint64_t i = 0; i < count; ++i) {
        uint64_t h = orc_splitmix64((seed ^ 0xA0761D6478BD642Full) + (first + i) * 0xE7037ED1A0B428DBull);
        w[i] = (float)((uint32_t)(h >> 40) + 1u) * (1.0f / 16777216.0f);
    }
}

/* ------------------------------------------------------------------------------------------
 * CSR construction — crates/builder/src/graph/csr.rs:124-221 (sequential semantics of the
 * degree / prefix-sum / scatter build), sort_targets :886-895, sort_and_deduplicate_targets
 * :897-948.  direction: 0 = Outgoing, 1 = Incoming, 2 = Undirected.
 * layout: 0 = Unsorted, 1 = Sorted, 2 = Deduplicated (csr.rs:34-45).
 * Values (weights) ride along; the reference orders Target by target only
 * (crates/builder/src/graph/mod.rs:20-36), equal targets keep a stable order here.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t t;
    float v;
    uint32_t seq;
} orc_tv;

static int orc_tv_cmp(const void *a, const void *b)
{
    const orc_tv *x = (const orc_tv *)a, *y = (const orc_tv *)b;
    if (x->t != y->t)
        return x->t < y->t ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

static int orc_u32_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

/* returns the resulting edge (target) count, or UINT64_MAX on allocation failure.
 * offsets_out: n+1, targets_out / weights_out: capacity m (2m for Undirected). */
ORC_EXPORT uint64_t orc_csr_build(uint32_t n, uint64_t m, const uint32_t *src, const uint32_t *dst,
                                  const float *w, int direction, int layout, uint32_t *offsets_out,
                                  uint32_t *targets_out, float *weights_out)
{
    uint64_t total = (direction == 2) ? 2 * m : m;
    uint32_t *cursor = (uint32_t *)calloc((size_t)n + 1, sizeof(uint32_t));
    if (!cursor)
        return UINT64_MAX;
    /* degrees (input/edgelist.rs:61-78) */
    for (uint64_t i = 0; i < m; ++i) {
        if (direction == 0 || direction == 2)
            cursor[src[i]]++;
        if (direction == 1 || direction == 2)
            cursor[dst[i]]++;
    }
    /* exclusive prefix sum (csr.rs:854-869) */
    uint32_t run = 0;
    for (uint32_t u = 0; u < n; ++u) {
        uint32_t d = cursor[u];
        offsets_out[u] = run;
        cursor[u] = run;
        run += d;
    }
    offsets_out[n] = run;
    /* scatter in edge-list order: all out-direction entries first, then in-direction (csr.rs:154-172) */
    if (direction == 0 || direction == 2)
        for (uint64_t i = 0; i < m; ++i) {
            uint32_t p = cursor[src[i]]++;
            targets_out[p] = dst[i];
            if (w)
                weights_out[p] = w[i];
        }
    if (direction == 1 || direction == 2)
        for (uint64_t i = 0; i < m; ++i) {
            uint32_t p = cursor[dst[i]]++;
            targets_out[p] = src[i];
            if (w)
                weights_out[p] = w[i];
        }
    free(cursor);
    if (layout == 0)
        return total;

    /* per-list sort (stable w.r.t. arrival order for equal targets) */
    uint32_t maxdeg = 0;
    for (uint32_t u = 0; u < n; ++u) {
        uint32_t d = offsets_out[u + 1] - offsets_out[u];
        if (d > maxdeg)
            maxdeg = d;
    }
    orc_tv *tmp = NULL;
    if (w) {
        tmp = (orc_tv *)malloc(((size_t)maxdeg + 1) * sizeof(orc_tv));
        if (!tmp)
            return UINT64_MAX;
    }
    for (uint32_t u = 0; u < n; ++u) {
        uint32_t s = offsets_out[u], e = offsets_out[u + 1];
        if (e - s < 2)
            continue;
        if (w) {
            for (uint32_t i = s; i < e; ++i) {
                tmp[i - s].t = targets_out[i];
                tmp[i - s].v = weights_out[i];
                tmp[i - s].seq = i - s;
            }
            qsort(tmp, e - s, sizeof(orc_tv), orc_tv_cmp);
            for (uint32_t i = s; i < e; ++i) {
                targets_out[i] = tmp[i - s].t;
                weights_out[i] = tmp[i - s].v;
            }
        } else {
            qsort(targets_out + s, e - s, sizeof(uint32_t), orc_u32_cmp);
        }
    }
    free(tmp);
    if (layout == 1)
        return total;

    /* Deduplicated: drop equal-target duplicates (first of each run survives), drop the
     * self-loop entry (csr.rs:907-922), compact. */
    uint64_t wr = 0;
    uint32_t rd_begin = 0;
    for (uint32_t u = 0; u < n; ++u) {
        uint32_t s = rd_begin, e = offsets_out[u + 1];
        rd_begin = e;
        offsets_out[u] = (uint32_t)wr;
        uint32_t prev = 0;
        for (uint32_t i = s; i < e; ++i) {
            uint32_t t = targets_out[i];
            int dup = (i > s && prev == t);
            prev = t;
            if (dup || t == u)
                continue;
            targets_out[wr] = t; /* wr <= i: in-place compaction never overtakes the read cursor */
            if (w)
                weights_out[wr] = weights_out[i];
            ++wr;
        }
    }
    offsets_out[n] = (uint32_t)wr;
    return wr;
}

/* ------------------------------------------------------------------------------------------
 * Relabel by degree — crates/builder/src/graph_ops.rs:511-638 (make_degree_ordered).
 * pairs (degree, node) sorted DESCENDING lexicographically (:555, ties: larger old id gets
 * the smaller new id); new_id[old] = rank (:564-592); lists relabelled then sorted (:606-630).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t deg, node;
} orc_dn;

static int orc_dn_cmp_desc(const void *a, const void *b)
{
    const orc_dn *x = (const orc_dn *)a, *y = (const orc_dn *)b;
    if (x->deg != y->deg)
        return x->deg > y->deg ? -1 : 1;
    if (x->node != y->node)
        return x->node > y->node ? -1 : 1;
    return 0;
}

ORC_EXPORT int orc_relabel_by_degree(uint32_t n, const uint32_t *off, const uint32_t *tgt,
                                     uint32_t *new_off, uint32_t *new_tgt, uint32_t *new_id)
{
    orc_dn *pairs = (orc_dn *)malloc((size_t)(n ? n : 1) * sizeof(orc_dn));
    if (!pairs)
        return -1;
    for (uint32_t u = 0; u < n; ++u) {
        pairs[u].deg = off[u + 1] - off[u];
        pairs[u].node = u;
    }
    qsort(pairs, n, sizeof(orc_dn), orc_dn_cmp_desc);
    uint32_t run = 0;
    for (uint32_t k = 0; k < n; ++k) {
        new_id[pairs[k].node] = k;
        new_off[k] = run;
        run += pairs[k].deg;
    }
    new_off[n] = run;
    for (uint32_t u = 0; u < n; ++u) {
        uint32_t nu = new_id[u], p = new_off[nu];
        for (uint32_t i = off[u]; i < off[u + 1]; ++i)
            new_tgt[p++] = new_id[tgt[i]];
        qsort(new_tgt + new_off[nu], p - new_off[nu], sizeof(uint32_t), orc_u32_cmp);
    }
    free(pairs);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PageRank — crates/algos/src/page_rank.rs:58-168.
 *   init (:70-81): score = 1/n, out_score = (1/n)/out_degree (f32; /0 -> +inf, never read)
 *   sweep (:142-160): incoming = f32 sum in CSR order of out_scores[v]; new = base + d*incoming;
 *                     error += |new - old| (diff in f32, widened to f64); out_scores[u] = new/out_deg
 *                     written IN PLACE (visible to later nodes of the same sweep).
 *   stop (:105-109): iteration += 1; if error < tolerance || iteration == max_iterations.
 * orc_page_rank_seq: one thread, ascending u — what the reference does for n <= 16384 (one
 * chunk) or on one hardware thread; bit-exact with the reference's golden vectors.
 * ------------------------------------------------------------------------------------------ */
ORC_EXPORT void orc_page_rank_seq(uint32_t n, const uint32_t *in_off, const uint32_t *in_tgt,
                                  const uint32_t *out_deg, uint64_t max_iterations, double tolerance,
                                  float damping, float *scores, uint64_t *iterations_out, double *error_out)
{
    const float init = 1.0f / (float)n;
    const float base = (1.0f - damping) / (float)n;
    float *outs = (float *)malloc((size_t)(n ? n : 1) * sizeof(float));
    for (uint32_t u = 0; u < n; ++u) {
        scores[u] = init;
        outs[u] = init / (float)out_deg[u];
    }
    uint64_t iter = 0;
    double err = 0.0;
    for (;;) {
        err = 0.0;
        for (uint32_t u = 0; u < n; ++u) {
            float s = 0.0f;
            for (uint32_t i = in_off[u]; i < in_off[u + 1]; ++i)
                s = s + outs[in_tgt[i]];
            float old = scores[u];
            float prod = damping * s;
            float nw = base + prod;
            scores[u] = nw;
            float diff = nw - old;
            err += fabs((double)diff);
            outs[u] = nw / (float)out_deg[u];
        }
        iter += 1;
        if (err < tolerance || iter == max_iterations)
            break;
    }
    free(outs);
    *iterations_out = iter;
    *error_out = err;
}

/* The multi-threaded reference path (:127-165): T scoped threads re-spawned every sweep pull
 * 16384-node chunks from an atomic cursor; out_scores raced in place; per-thread f64 error
 * added to one atomic.  This is what bench.py times as cpu_baseline ("port"). */
#define ORC_PR_CHUNK 16384u

typedef struct {
    uint32_t n;
    const uint32_t *in_off, *in_tgt, *out_deg;
    float base, damping;
    float *scores;
    float *outs;
    atomic_uint_fast64_t *next_chunk;
    double err;
} orc_pr_job;

static void *orc_pr_worker(void *arg)
{
    orc_pr_job *j = (orc_pr_job *)arg;
    double err = 0.0;
    for (;;) {
        uint64_t start = atomic_fetch_add(j->next_chunk, ORC_PR_CHUNK);
        if (start >= j->n)
            break;
        uint64_t end = start + ORC_PR_CHUNK;
        if (end > j->n)
            end = j->n;
        for (uint32_t u = (uint32_t)start; u < (uint32_t)end; ++u) {
            float s = 0.0f;
            for (uint32_t i = j->in_off[u]; i < j->in_off[u + 1]; ++i)
                s = s + ((volatile float *)j->outs)[j->in_tgt[i]];
            float old = j->scores[u];
            float prod = j->damping * s;
            float nw = j->base + prod;
            j->scores[u] = nw;
            float diff = nw - old;
            err += fabs((double)diff);
            ((volatile float *)j->outs)[u] = nw / (float)j->out_deg[u];
        }
    }
    j->err = err;
    return NULL;
}

/* The same loop for the TIMED baseline only (orc_page_rank_chunked_timed): plain loads and stores of
 * out_scores, as the reference's raw-pointer SharedMut accesses compile to (page_rank.rs:143-159) —
 * the checker above keeps `volatile` so that the racing in-place updates it models are really re-read. */
static void *orc_pr_worker_timed(void *arg)
{
    orc_pr_job *j = (orc_pr_job *)arg;
    const uint32_t *restrict in_off = j->in_off, *restrict in_tgt = j->in_tgt, *restrict out_deg = j->out_deg;
    float *outs = j->outs, *scores = j->scores;
    const float base = j->base, damping = j->damping;
    double err = 0.0;
    for (;;) {
        uint64_t start = atomic_fetch_add(j->next_chunk, ORC_PR_CHUNK);
        if (start >= j->n)
            break;
        uint64_t end = start + ORC_PR_CHUNK;
        if (end > j->n)
            end = j->n;
        for (uint32_t u = (uint32_t)start; u < (uint32_t)end; ++u) {
            float s = 0.0f;
            for (uint32_t i = in_off[u]; i < in_off[u + 1]; ++i)
                s = s + outs[in_tgt[i]];
            const float old = scores[u];
            const float prod = damping * s;
            const float nw = base + prod;
            scores[u] = nw;
            err += fabs((double)(nw - old));
            outs[u] = nw / (float)out_deg[u];
        }
    }
    j->err = err;
    return NULL;
}

/* seconds_per_iter_out (optional, may be NULL): wall time of every sweep, length max_iterations */
ORC_EXPORT int orc_page_rank_chunked(uint32_t n, const uint32_t *in_off, const uint32_t *in_tgt,
                                     const uint32_t *out_deg, uint64_t max_iterations, double tolerance,
                                     float damping, uint32_t threads, float *scores,
                                     uint64_t *iterations_out, double *error_out)
{
    if (threads == 0)
        threads = 4; /* DEFAULT_PARALLELISM, crates/algos/src/lib.rs:152 */
    const float init = 1.0f / (float)n;
    const float base = (1.0f - damping) / (float)n;
    float *outs = (float *)malloc((size_t)(n ? n : 1) * sizeof(float));
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_pr_job *jobs = (orc_pr_job *)malloc(threads * sizeof(orc_pr_job));
    if (!outs || !tid || !jobs)
        return -1;
    for (uint32_t u = 0; u < n; ++u) {
        scores[u] = init;
        outs[u] = init / (float)out_deg[u];
    }
    uint64_t iter = 0;
    double err = 0.0;
    for (;;) {
        atomic_uint_fast64_t next;
        atomic_init(&next, 0);
        for (uint32_t t = 0; t < threads; ++t) {
            jobs[t] = (orc_pr_job){n, in_off, in_tgt, out_deg, base, damping, scores, outs, &next, 0.0};
            pthread_create(&tid[t], NULL, orc_pr_worker, &jobs[t]);
        }
        err = 0.0;
        for (uint32_t t = 0; t < threads; ++t) {
            pthread_join(tid[t], NULL);
            err += jobs[t].err;
        }
        iter += 1;
        if (err < tolerance || iter == max_iterations)
            break;
    }
    free(outs);
    free(tid);
    free(jobs);
    *iterations_out = iter;
    *error_out = err;
    return 0;
}

/* ---- timed CPU baseline (bench.py's cpu_baseline leg) ------------------------------------------------
 * The same sweeps as orc_page_rank_chunked on private copies of the inputs whose pages were first
 * touched in 2 MiB stripes by all worker threads.  The reference builds its CSR with the rayon pool
 * (csr.rs:124-221), so its pages are spread over the NUMA nodes of a many-socket host; arrays handed
 * over from numpy were touched by one thread and would make every core pull from one memory
 * controller.  spread = 0 runs on the caller's arrays as they are.  Returns the wall time of `sweeps`
 * sweeps after one untimed sweep; thread creation per sweep is inside the time, as in the reference
 * (page_rank.rs:127-131). */
typedef struct {
    char *dst;
    const char *src; /* NULL: fill with zero bytes */
    size_t bytes;
    uint32_t t, T;
} orc_spread_job;

static void *orc_spread_worker(void *p)
{
    orc_spread_job *j = (orc_spread_job *)p;
    const size_t stripe = (size_t)2 << 20;
    for (size_t o = (size_t)j->t * stripe; o < j->bytes; o += (size_t)j->T * stripe) {
        const size_t len = j->bytes - o < stripe ? j->bytes - o : stripe;
        if (j->src)
            memcpy(j->dst + o, j->src + o, len);
        else
            memset(j->dst + o, 0, len);
    }
    return NULL;
}

static void *orc_spread_copy(const void *src, size_t bytes, uint32_t threads)
{
    /* 2 MiB-aligned and advised as huge pages: the gathered vector spans 65536 4-KiB pages at scale 26,
     * far beyond the TLB reach (a system with transparent_hugepage=always gives the reference the same) */
    void *mem = NULL;
    if (posix_memalign(&mem, (size_t)2 << 20, bytes ? bytes : 1) != 0)
        return NULL;
    (void)madvise(mem, bytes, MADV_HUGEPAGE);
    char *dst = (char *)mem;
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_spread_job *jobs = (orc_spread_job *)malloc(threads * sizeof(orc_spread_job));
    for (uint32_t t = 0; t < threads; ++t) {
        jobs[t] = (orc_spread_job){dst, (const char *)src, bytes, t, threads};
        pthread_create(&tid[t], NULL, orc_spread_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; ++t)
        pthread_join(tid[t], NULL);
    free(tid);
    free(jobs);
    return dst;
}

ORC_EXPORT int orc_page_rank_chunked_timed(uint32_t n, const uint32_t *in_off, const uint32_t *in_tgt,
                                           const uint32_t *out_deg, uint64_t sweeps, float damping, uint32_t threads,
                                           int spread, double *seconds_out, double *error_out)
{
    if (threads == 0)
        threads = 4;
    const uint64_t m = in_off[n];
    const uint32_t *off = in_off, *tgt = in_tgt, *od = out_deg;
    void *c_off = NULL, *c_tgt = NULL, *c_od = NULL;
    if (spread) {
        c_off = orc_spread_copy(in_off, ((size_t)n + 1) * 4, threads);
        c_tgt = orc_spread_copy(in_tgt, (size_t)m * 4, threads);
        c_od = orc_spread_copy(out_deg, (size_t)n * 4, threads);
        if (!c_off || !c_tgt || !c_od)
            return -1;
        off = (const uint32_t *)c_off, tgt = (const uint32_t *)c_tgt, od = (const uint32_t *)c_od;
    }
    float *scores = (float *)(spread ? orc_spread_copy(NULL, (size_t)n * 4, threads) : malloc((size_t)(n ? n : 1) * 4));
    float *outs = (float *)(spread ? orc_spread_copy(NULL, (size_t)n * 4, threads) : malloc((size_t)(n ? n : 1) * 4));
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_pr_job *jobs = (orc_pr_job *)malloc(threads * sizeof(orc_pr_job));
    if (!scores || !outs || !tid || !jobs)
        return -1;
    const float init = 1.0f / (float)n, base = (1.0f - damping) / (float)n;
    for (uint32_t u = 0; u < n; ++u) {
        scores[u] = init;
        outs[u] = init / (float)od[u];
    }
    struct timespec t0, t1;
    double err = 0.0;
    for (uint64_t it = 0; it <= sweeps; ++it) {
        if (it == 1)
            clock_gettime(CLOCK_MONOTONIC, &t0);
        atomic_uint_fast64_t next;
        atomic_init(&next, 0);
        for (uint32_t t = 0; t < threads; ++t) {
            jobs[t] = (orc_pr_job){n, off, tgt, od, base, damping, scores, outs, &next, 0.0};
            pthread_create(&tid[t], NULL, orc_pr_worker_timed, &jobs[t]);
        }
        err = 0.0;
        for (uint32_t t = 0; t < threads; ++t) {
            pthread_join(tid[t], NULL);
            err += jobs[t].err;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    *error_out = err;
    free(scores);
    free(outs);
    free(tid);
    free(jobs);
    free(c_off);
    free(c_tgt);
    free(c_od);
    return 0;
}

/* f64 Jacobi fixed-point reference (not in the reference; used to judge which of two f32
 * implementations is closer to the exact fixed point of the reference's recurrence). */
ORC_EXPORT void orc_page_rank_f64(uint32_t n, const uint32_t *in_off, const uint32_t *in_tgt,
                                  const uint32_t *out_deg, uint64_t max_iterations, double tolerance,
                                  double damping, double *scores, uint64_t *iterations_out, double *error_out)
{
    const double init = 1.0 / (double)n, base = (1.0 - damping) / (double)n;
    double *outs = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    for (uint32_t u = 0; u < n; ++u) {
        scores[u] = init;
        outs[u] = out_deg[u] ? init / (double)out_deg[u] : 0.0;
    }
    uint64_t iter = 0;
    double err = 0.0;
    for (;;) {
        err = 0.0;
        for (uint32_t u = 0; u < n; ++u) {
            double s = 0.0;
            for (uint32_t i = in_off[u]; i < in_off[u + 1]; ++i)
                s += outs[in_tgt[i]];
            double nw = base + damping * s;
            err += fabs(nw - scores[u]);
            scores[u] = nw;
            outs[u] = out_deg[u] ? nw / (double)out_deg[u] : 0.0; /* Gauss-Seidel: same fixed point */
        }
        iter += 1;
        if (err < tolerance || iter == max_iterations)
            break;
    }
    free(outs);
    *iterations_out = iter;
    *error_out = err;
}

/* One synchronous (Jacobi) f32 sweep with the reference's per-node arithmetic but reading a
 * frozen copy of out_scores — the order-free definition the GPU kernel implements. Sums in CSR
 * order; used on tiny inputs to check the kernel's arithmetic other than summation order. */
ORC_EXPORT double orc_page_rank_jacobi_sweep(uint32_t n, const uint32_t *in_off, const uint32_t *in_tgt,
                                             const uint32_t *out_deg, float damping, float *scores,
                                             const float *outs_in, float *outs_out)
{
    const float base = (1.0f - damping) / (float)n;
    double err = 0.0;
    for (uint32_t u = 0; u < n; ++u) {
        float s = 0.0f;
        for (uint32_t i = in_off[u]; i < in_off[u + 1]; ++i)
            s = s + outs_in[in_tgt[i]];
        float prod = damping * s;
        float nw = base + prod;
        float diff = nw - scores[u];
        err += fabs((double)diff);
        scores[u] = nw;
        outs_out[u] = nw / (float)out_deg[u];
    }
    return err;
}

/* ------------------------------------------------------------------------------------------
 * Afforest union-find — crates/algos/src/afforest.rs:22-56 (single-thread restatement; the
 * CAS at :33 always succeeds when its precondition holds).
 * ------------------------------------------------------------------------------------------ */
static void orc_af_union(uint32_t *parent, uint32_t u, uint32_t v)
{
    uint32_t p1 = parent[u], p2 = parent[v];
    while (p1 != p2) {
        uint32_t high = p1 > p2 ? p1 : p2;
        uint32_t low = p1 + p2 - high;
        uint32_t p_high = parent[high];
        if (p_high == low)
            break;
        if (p_high == high) {
            parent[high] = low;
            break;
        }
        p1 = parent[parent[high]];
        p2 = parent[low];
    }
}

static void orc_af_compress(uint32_t *parent, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
        while (parent[i] != parent[parent[i]])
            parent[i] = parent[parent[i]];
}

/* Disjoint-set struct — crates/algos/src/dss.rs:38-116: find with path halving (:76-94),
 * union-by-min (:38-62), compress = find(id) for every id (:112-116). */
static uint32_t orc_dss_find(uint32_t *parent, uint32_t id)
{
    uint32_t p = parent[id];
    while (id != p) {
        uint32_t gp = parent[p];
        if (parent[id] == p)
            parent[id] = gp;
        id = p;
        p = gp;
    }
    return id;
}

static void orc_dss_union(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (;;) {
        a = orc_dss_find(parent, a);
        b = orc_dss_find(parent, b);
        if (a == b)
            return;
        if (a < b) {
            uint32_t t = a;
            a = b;
            b = t;
        }
        if (parent[a] == a) {
            parent[a] = b;
            return;
        }
    }
}

ORC_EXPORT void orc_uf_new(uint32_t n, uint32_t *parent)
{
    for (uint32_t i = 0; i < n; ++i)
        parent[i] = i;
}
/* kind: 0 = Afforest, 1 = DisjointSetStruct */
ORC_EXPORT void orc_uf_union(int kind, uint32_t *parent, uint32_t u, uint32_t v)
{
    if (kind == 0)
        orc_af_union(parent, u, v);
    else
        orc_dss_union(parent, u, v);
}
ORC_EXPORT uint32_t orc_uf_find(int kind, uint32_t *parent, uint32_t u)
{
    return kind == 0 ? parent[u] : orc_dss_find(parent, u);
}
ORC_EXPORT void orc_uf_compress(int kind, uint32_t *parent, uint32_t n)
{
    if (kind == 0)
        orc_af_compress(parent, n);
    else
        for (uint32_t i = 0; i < n; ++i)
            (void)orc_dss_find(parent, i);
}

/* ------------------------------------------------------------------------------------------
 * WCC — crates/algos/src/wcc.rs:103-301.
 * algo: 0 = wcc_afforest (:127), 1 = wcc_afforest_dss (:144), 2 = wcc_baseline (:103-122).
 * Pipeline (:164-182): sample_subgraph (:186-204) -> compress -> find_largest_component
 * (:245-271; the reference draws from an unseeded WyRand — which component is skipped never
 * changes the result; a seeded splitmix64 stands in) -> link_remaining (:274-301) -> compress.
 * components_out[u] = component(u) (== find(u), the root, for both backends; see dss.rs:154).
 * ------------------------------------------------------------------------------------------ */
ORC_EXPORT int orc_wcc(int algo, uint32_t n, const uint32_t *out_off, const uint32_t *out_tgt,
                       const uint32_t *in_off, const uint32_t *in_tgt, uint64_t neighbor_rounds,
                       uint64_t sampling_size, uint64_t seed, uint32_t *components_out)
{
    uint32_t *parent = components_out;
    orc_uf_new(n, parent);
    if (n == 0)
        return 0;
    if (algo == 2) {
        for (uint32_t u = 0; u < n; ++u)
            for (uint32_t i = out_off[u]; i < out_off[u + 1]; ++i)
                orc_dss_union(parent, u, out_tgt[i]);
        for (uint32_t u = 0; u < n; ++u)
            parent[u] = orc_dss_find(parent, u);
        return 0;
    }
    const int kind = (algo == 0) ? 0 : 1;
    if (sampling_size == 0)
        return -2; /* reference: unwrap() on an empty map panics (wcc.rs:260-263) */
    for (uint32_t u = 0; u < n; ++u) {
        uint64_t deg = out_off[u + 1] - out_off[u];
        uint64_t take = deg < neighbor_rounds ? deg : neighbor_rounds;
        for (uint64_t k = 0; k < take; ++k)
            orc_uf_union(kind, parent, u, out_tgt[out_off[u] + k]);
    }
    orc_uf_compress(kind, parent, n);
    /* most frequent sampled component; ties -> first maximal (reference: arbitrary) */
    uint32_t *samples = (uint32_t *)malloc(sampling_size * sizeof(uint32_t));
    if (!samples)
        return -1;
    for (uint64_t k = 0; k < sampling_size; ++k)
        samples[k] = orc_uf_find(kind, parent, (uint32_t)(orc_splitmix64(seed + k) % n));
    qsort(samples, sampling_size, sizeof(uint32_t), orc_u32_cmp);
    uint32_t skip = samples[0];
    uint64_t best = 0, runlen = 0;
    for (uint64_t k = 0; k < sampling_size; ++k) {
        runlen = (k > 0 && samples[k] == samples[k - 1]) ? runlen + 1 : 1;
        if (runlen > best) {
            best = runlen;
            skip = samples[k];
        }
    }
    free(samples);
    for (uint32_t u = 0; u < n; ++u) {
        if (orc_uf_find(kind, parent, u) == skip)
            continue;
        uint64_t deg = out_off[u + 1] - out_off[u];
        if (deg > neighbor_rounds)
            for (uint64_t i = out_off[u] + neighbor_rounds; i < out_off[u + 1]; ++i)
                orc_uf_union(kind, parent, u, out_tgt[i]);
        for (uint32_t i = in_off[u]; i < in_off[u + 1]; ++i)
            orc_uf_union(kind, parent, u, in_tgt[i]);
    }
    orc_uf_compress(kind, parent, n);
    if (kind == 1) /* Components::component(u) = find(u) (dss.rs:154-156) */
        for (uint32_t u = 0; u < n; ++u)
            parent[u] = orc_dss_find(parent, u);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Delta-stepping SSSP — crates/algos/src/sssp.rs:38-204, restated for one rayon thread
 * (T = 1: one ThreadLocalBins, the shared frontier is drained in 64-node batches in order).
 * INF = f32::MAX (:12); bucket of d = (usize)(d/delta) (:192); stale-entry check
 * dist >= delta*curr_bin (:126); own current bin re-drained while 0 < len < 1000 (:145-155).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t *v;
    size_t len, cap;
} orc_bin;

typedef struct {
    orc_bin *bins;
    size_t len, cap;
} orc_bins;

static void orc_bins_resize(orc_bins *b, size_t new_len)
{
    if (new_len > b->cap) {
        size_t nc = b->cap ? b->cap : 8;
        while (nc < new_len)
            nc *= 2;
        b->bins = (orc_bin *)realloc(b->bins, nc * sizeof(orc_bin));
        memset(b->bins + b->cap, 0, (nc - b->cap) * sizeof(orc_bin));
        b->cap = nc;
    }
    if (new_len > b->len)
        b->len = new_len;
}

static void orc_bin_push(orc_bin *b, uint32_t x)
{
    if (b->len == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 16;
        b->v = (uint32_t *)realloc(b->v, b->cap * sizeof(uint32_t));
    }
    b->v[b->len++] = x;
}

static void orc_relax_edges(const uint32_t *off, const uint32_t *tgt, const float *w, float *dist,
                            orc_bins *bins, uint32_t node, float delta)
{
    for (uint32_t i = off[node]; i < off[node + 1]; ++i) {
        uint32_t t = tgt[i];
        float old_d = dist[t];
        float new_d = dist[node] + w[i];
        if (new_d < old_d) {
            dist[t] = new_d;
            size_t dest = (size_t)(new_d / delta);
            if (dest >= bins->len)
                orc_bins_resize(bins, dest + 1);
            orc_bin_push(&bins->bins[dest], t);
        }
    }
}

ORC_EXPORT int orc_delta_stepping(uint32_t n, const uint32_t *off, const uint32_t *tgt, const float *w,
                                  uint64_t start_node, float delta, float *dist)
{
    if (start_node >= n)
        return -2; /* reference: index out of bounds panic (sssp.rs:52) */
    for (uint32_t i = 0; i < n; ++i)
        dist[i] = FLT_MAX;
    dist[start_node] = 0.0f;
    size_t fcap = 64, flen = 1;
    uint32_t *frontier = (uint32_t *)malloc(fcap * sizeof(uint32_t));
    frontier[0] = (uint32_t)start_node;
    orc_bins bins = {0};
    orc_bins_resize(&bins, 1);
    size_t curr = 0;
    const size_t NO_BIN = (size_t)-1;
    while (curr != NO_BIN) {
        /* process_shared_bin (:104-132) */
        for (size_t k = 0; k < flen; ++k) {
            uint32_t node = frontier[k];
            if (dist[node] >= delta * (float)curr)
                orc_relax_edges(off, tgt, w, dist, &bins, node, delta);
        }
        /* process_local_bins (:134-157) */
        while (curr < bins.len && bins.bins[curr].len != 0 && bins.bins[curr].len < 1000) {
            size_t cl = bins.bins[curr].len;
            uint32_t *copy = (uint32_t *)malloc(cl * sizeof(uint32_t));
            memcpy(copy, bins.bins[curr].v, cl * sizeof(uint32_t));
            bins.bins[curr].len = 0;
            for (size_t k = 0; k < cl; ++k)
                orc_relax_edges(off, tgt, w, dist, &bins, copy[k], delta);
            free(copy);
        }
        /* min_non_empty_bin (:159-168) */
        size_t next = NO_BIN;
        for (size_t b = curr; b < bins.len; ++b)
            if (bins.bins[b].len != 0) {
                next = b;
                break;
            }
        /* copy next local bin into the shared frontier (:85-94) */
        flen = 0;
        if (next != NO_BIN) {
            flen = bins.bins[next].len;
            if (flen > fcap) {
                fcap = flen;
                frontier = (uint32_t *)realloc(frontier, fcap * sizeof(uint32_t));
            }
            memcpy(frontier, bins.bins[next].v, flen * sizeof(uint32_t));
            bins.bins[next].len = 0;
        }
        curr = next;
    }
    for (size_t b = 0; b < bins.cap; ++b)
        free(bins.bins[b].v);
    free(bins.bins);
    free(frontier);
    return 0;
}

/* ---- timed CPU baselines of WCC and delta-stepping (tools/bench_algos.py's cpu_baseline legs) ------------------------
 * The reference runs both on the rayon pool: wcc.rs:186-301 (`into_par_iter().chunks(chunk_size)` over the nodes,
 * Afforest::union with a CAS, afforest.rs:22-39, a parallel compress, :47-53) and sssp.rs:64-94 (one ThreadLocalBins per
 * thread, the shared frontier drained in 64-node batches from an atomic cursor, CAS on the distances, :170-204).  These
 * two functions restate that threading with pthreads for the TIMED leg only; the checkers above stay sequential.  Both
 * return what the sequential functions return on the inputs they are used on (labels are schedule-free; distances are
 * schedule-free unless the reference's stale check misfires, see orc_sssp_fixed_point). */
typedef struct {
    pthread_barrier_t *bar;
    uint32_t t, T, n;
    const uint32_t *out_off, *out_tgt, *in_off, *in_tgt;
    uint32_t *parent;
    uint64_t neighbor_rounds, chunk;
    atomic_uint_fast64_t *cursor; /* [4]: one per parallel phase */
    const uint32_t *skip;         /* set by thread 0 between the phases */
} orc_wcc_job;

static inline uint32_t orc_ld(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

static void orc_af_union_mt(uint32_t *parent, uint32_t u, uint32_t v)
{
    uint32_t p1 = orc_ld(&parent[u]), p2 = orc_ld(&parent[v]);
    while (p1 != p2) {
        uint32_t high = p1 > p2 ? p1 : p2;
        uint32_t low = p1 + p2 - high;
        uint32_t p_high = orc_ld(&parent[high]);
        if (p_high == low)
            break;
        if (p_high == high) {
            uint32_t expect = high;
            if (__atomic_compare_exchange_n(&parent[high], &expect, low, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
                break;
        }
        p1 = orc_ld(&parent[orc_ld(&parent[high])]);
        p2 = orc_ld(&parent[low]);
    }
}

static void orc_wcc_compress_mt(orc_wcc_job *j, atomic_uint_fast64_t *cursor)
{
    for (;;) {
        uint64_t s = atomic_fetch_add(cursor, 65536);
        if (s >= j->n)
            break;
        uint64_t e = s + 65536 < j->n ? s + 65536 : j->n;
        for (uint64_t i = s; i < e; ++i)
            while (orc_ld(&j->parent[i]) != orc_ld(&j->parent[orc_ld(&j->parent[i])]))
                __atomic_store_n(&j->parent[i], orc_ld(&j->parent[orc_ld(&j->parent[i])]), __ATOMIC_SEQ_CST);
    }
}

static void *orc_wcc_worker(void *arg)
{
    orc_wcc_job *j = (orc_wcc_job *)arg;
    /* sample_subgraph (wcc.rs:186-204) */
    for (;;) {
        uint64_t s = atomic_fetch_add(&j->cursor[0], j->chunk);
        if (s >= j->n)
            break;
        uint64_t e = s + j->chunk < j->n ? s + j->chunk : j->n;
        for (uint64_t u = s; u < e; ++u) {
            uint64_t deg = j->out_off[u + 1] - j->out_off[u];
            uint64_t take = deg < j->neighbor_rounds ? deg : j->neighbor_rounds;
            for (uint64_t k = 0; k < take; ++k)
                orc_af_union_mt(j->parent, (uint32_t)u, j->out_tgt[j->out_off[u] + k]);
        }
    }
    pthread_barrier_wait(j->bar);
    orc_wcc_compress_mt(j, &j->cursor[1]);
    pthread_barrier_wait(j->bar); /* thread 0 samples the largest component (sequential in the reference too) */
    pthread_barrier_wait(j->bar);
    const uint32_t skip = *j->skip;
    /* link_remaining (wcc.rs:274-301) */
    for (;;) {
        uint64_t s = atomic_fetch_add(&j->cursor[2], j->chunk);
        if (s >= j->n)
            break;
        uint64_t e = s + j->chunk < j->n ? s + j->chunk : j->n;
        for (uint64_t u = s; u < e; ++u) {
            if (orc_ld(&j->parent[u]) == skip)
                continue;
            uint64_t deg = j->out_off[u + 1] - j->out_off[u];
            if (deg > j->neighbor_rounds)
                for (uint64_t i = j->out_off[u] + j->neighbor_rounds; i < j->out_off[u + 1]; ++i)
                    orc_af_union_mt(j->parent, (uint32_t)u, j->out_tgt[i]);
            for (uint32_t i = j->in_off[u]; i < j->in_off[u + 1]; ++i)
                orc_af_union_mt(j->parent, (uint32_t)u, j->in_tgt[i]);
        }
    }
    pthread_barrier_wait(j->bar);
    orc_wcc_compress_mt(j, &j->cursor[3]);
    return NULL;
}

/* wcc_afforest on `threads` threads; seconds_out = wall time of the five phases (the union-find's creation excluded, as
 * the reference logs it separately, wcc.rs:132-141) */
ORC_EXPORT int orc_wcc_afforest_timed(uint32_t n, const uint32_t *out_off, const uint32_t *out_tgt, const uint32_t *in_off,
                                      const uint32_t *in_tgt, uint64_t neighbor_rounds, uint64_t sampling_size, uint64_t seed,
                                      uint64_t chunk_size, uint32_t threads, uint32_t *components_out, double *seconds_out)
{
    if (threads == 0)
        threads = 4;
    if (n == 0 || sampling_size == 0 || chunk_size == 0)
        return -2;
    uint32_t *parent = components_out;
    orc_uf_new(n, parent);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads + 1);
    atomic_uint_fast64_t cursor[4];
    for (int k = 0; k < 4; ++k)
        atomic_init(&cursor[k], 0);
    uint32_t skip = 0;
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_wcc_job *jobs = (orc_wcc_job *)malloc(threads * sizeof(orc_wcc_job));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t t = 0; t < threads; ++t) {
        jobs[t] = (orc_wcc_job){&bar, t, threads, n, out_off, out_tgt, in_off, in_tgt, parent, neighbor_rounds, chunk_size, cursor, &skip};
        pthread_create(&tid[t], NULL, orc_wcc_worker, &jobs[t]);
    }
    pthread_barrier_wait(&bar); /* subgraph linked */
    pthread_barrier_wait(&bar); /* compressed */
    {
        uint32_t *samples = (uint32_t *)malloc(sampling_size * sizeof(uint32_t));
        for (uint64_t k = 0; k < sampling_size; ++k)
            samples[k] = parent[orc_splitmix64(seed + k) % n];
        qsort(samples, sampling_size, sizeof(uint32_t), orc_u32_cmp);
        uint64_t best = 0, runlen = 0;
        skip = samples[0];
        for (uint64_t k = 0; k < sampling_size; ++k) {
            runlen = (k > 0 && samples[k] == samples[k - 1]) ? runlen + 1 : 1;
            if (runlen > best) {
                best = runlen;
                skip = samples[k];
            }
        }
        free(samples);
    }
    pthread_barrier_wait(&bar); /* skip component known */
    pthread_barrier_wait(&bar); /* remaining edges linked */
    for (uint32_t t = 0; t < threads; ++t)
        pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    pthread_barrier_destroy(&bar);
    free(tid);
    free(jobs);
    return 0;
}

typedef struct {
    pthread_barrier_t *bar;
    uint32_t t, T, n;
    const uint32_t *off, *tgt;
    const float *w;
    uint32_t *dist; /* f32 bit patterns (non-negative: ordered like the floats) */
    float delta;
    orc_bins bins;
    uint32_t *frontier;
    atomic_uint_fast64_t *cursor;
    size_t *curr, *flen;       /* shared, written by thread 0 between barriers */
    size_t *next_of, *len_of;  /* [T]: every thread's minimum non-empty bin / length of its `next` bin */
} orc_ds_job;

static void orc_relax_edges_mt(orc_ds_job *j, uint32_t node)
{
    for (uint32_t i = j->off[node]; i < j->off[node + 1]; ++i) {
        const uint32_t t = j->tgt[i];
        uint32_t old_bits = __atomic_load_n(&j->dist[t], __ATOMIC_ACQUIRE);
        uint32_t node_bits = __atomic_load_n(&j->dist[node], __ATOMIC_ACQUIRE);
        float node_d, old_d;
        memcpy(&node_d, &node_bits, 4);
        const float new_d = node_d + j->w[i];
        uint32_t new_bits;
        memcpy(&new_bits, &new_d, 4);
        memcpy(&old_d, &old_bits, 4);
        while (new_d < old_d) {
            if (__atomic_compare_exchange_n(&j->dist[t], &old_bits, new_bits, 1, __ATOMIC_RELEASE, __ATOMIC_RELAXED)) {
                size_t dest = (size_t)(new_d / j->delta);
                if (dest >= j->bins.len)
                    orc_bins_resize(&j->bins, dest + 1);
                orc_bin_push(&j->bins.bins[dest], t);
                break;
            }
            memcpy(&old_d, &old_bits, 4);
        }
    }
}

static void *orc_ds_worker(void *arg)
{
    orc_ds_job *j = (orc_ds_job *)arg;
    const size_t NO_BIN = (size_t)-1;
    for (;;) {
        pthread_barrier_wait(j->bar); /* curr / flen / cursor published */
        const size_t curr = *j->curr, flen = *j->flen;
        if (curr == NO_BIN)
            break;
        for (;;) { /* process_shared_bin (sssp.rs:104-132) */
            uint64_t o = atomic_fetch_add(j->cursor, 64);
            if (o >= flen)
                break;
            uint64_t lim = o + 64 < flen ? o + 64 : flen;
            for (uint64_t k = o; k < lim; ++k) {
                const uint32_t node = j->frontier[k];
                uint32_t b = __atomic_load_n(&j->dist[node], __ATOMIC_ACQUIRE);
                float d;
                memcpy(&d, &b, 4);
                if (d >= j->delta * (float)curr)
                    orc_relax_edges_mt(j, node);
            }
        }
        while (curr < j->bins.len && j->bins.bins[curr].len != 0 && j->bins.bins[curr].len < 1000) { /* :134-157 */
            size_t cl = j->bins.bins[curr].len;
            uint32_t *copy = (uint32_t *)malloc(cl * sizeof(uint32_t));
            memcpy(copy, j->bins.bins[curr].v, cl * sizeof(uint32_t));
            j->bins.bins[curr].len = 0;
            for (size_t k = 0; k < cl; ++k)
                orc_relax_edges_mt(j, copy[k]);
            free(copy);
        }
        size_t next = NO_BIN; /* :159-168 */
        for (size_t b = curr; b < j->bins.len; ++b)
            if (j->bins.bins[b].len != 0) {
                next = b;
                break;
            }
        j->next_of[j->t] = next;
        pthread_barrier_wait(j->bar); /* every thread's minimum known */
        size_t gnext = NO_BIN;
        for (uint32_t t = 0; t < j->T; ++t)
            gnext = j->next_of[t] < gnext ? j->next_of[t] : gnext;
        j->len_of[j->t] = (gnext != NO_BIN && gnext < j->bins.len) ? j->bins.bins[gnext].len : 0;
        pthread_barrier_wait(j->bar); /* every thread's slice length known (frontier_slices, :206-225) */
        if (gnext != NO_BIN) {
            size_t at = 0;
            for (uint32_t t = 0; t < j->t; ++t)
                at += j->len_of[t];
            if (j->len_of[j->t]) {
                memcpy(j->frontier + at, j->bins.bins[gnext].v, j->len_of[j->t] * sizeof(uint32_t));
                j->bins.bins[gnext].len = 0;
            }
        }
        pthread_barrier_wait(j->bar); /* frontier complete */
        if (j->t == 0) {
            size_t total = 0;
            for (uint32_t t = 0; t < j->T; ++t)
                total += j->len_of[t];
            *j->curr = gnext;
            *j->flen = total;
            atomic_store(j->cursor, 0);
        }
    }
    for (size_t b = 0; b < j->bins.cap; ++b)
        free(j->bins.bins[b].v);
    free(j->bins.bins);
    return NULL;
}

/* delta_stepping on `threads` persistent threads (the rayon pool); seconds_out = wall time from the first round on */
ORC_EXPORT int orc_delta_stepping_timed(uint32_t n, const uint32_t *off, const uint32_t *tgt, const float *w, uint64_t start_node,
                                        float delta, uint32_t threads, float *dist, double *seconds_out)
{
    if (start_node >= n)
        return -2;
    if (threads == 0)
        threads = 4;
    const float inf = FLT_MAX;
    uint32_t inf_bits;
    memcpy(&inf_bits, &inf, 4);
    uint32_t *bits = (uint32_t *)dist;
    for (uint32_t i = 0; i < n; ++i)
        bits[i] = inf_bits;
    bits[start_node] = 0;
    uint32_t *frontier = (uint32_t *)malloc(((size_t)off[n] + 1) * sizeof(uint32_t)); /* edge_count entries, sssp.rs:55 */
    if (!frontier)
        return -1;
    frontier[0] = (uint32_t)start_node;
    size_t curr = 0, flen = 1;
    atomic_uint_fast64_t cursor;
    atomic_init(&cursor, 0);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads);
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_ds_job *jobs = (orc_ds_job *)calloc(threads, sizeof(orc_ds_job));
    size_t *next_of = (size_t *)malloc(threads * sizeof(size_t)), *len_of = (size_t *)malloc(threads * sizeof(size_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t t = 0; t < threads; ++t) {
        jobs[t] = (orc_ds_job){&bar, t, threads, n, off, tgt, w, bits, delta, {0}, frontier, &cursor, &curr, &flen, next_of, len_of};
        orc_bins_resize(&jobs[t].bins, 1);
        pthread_create(&tid[t], NULL, orc_ds_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; ++t)
        pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    pthread_barrier_destroy(&bar);
    free(tid);
    free(jobs);
    free(next_of);
    free(len_of);
    free(frontier);
    return 0;
}

/* The least fixed point of d[v] = min(d[u] (+) w(u,v)) under f32 addition (f32 Dijkstra with a binary
 * heap): what delta-stepping computes under ANY schedule as long as no improvement is dropped.  The
 * reference does drop some: a node whose new distance d lands in bin (usize)(d/delta) (sssp.rs:192) is
 * skipped as "stale" when its turn comes if d < delta * bin (sssp.rs:126), and f32 rounding makes that
 * happen for real distances (d = 13.5, delta = 0.3f: 13.5/0.3f rounds up to 45.0, 0.3f*45 = 13.500001):
 * the node's edges are then never relaxed and everything behind it keeps a longer distance or f32::MAX.
 * orc_delta_stepping above restates that faithfully; this function is the intended result. */
typedef struct {
    float d;
    uint32_t v;
} orc_heap_item;

ORC_EXPORT int orc_sssp_fixed_point(uint32_t n, const uint32_t *off, const uint32_t *tgt, const float *w,
                                    uint64_t start_node, float *dist)
{
    if (start_node >= n)
        return -1;
    for (uint32_t u = 0; u < n; ++u)
        dist[u] = FLT_MAX;
    dist[start_node] = 0.0f;
    size_t cap = 1024, len = 0;
    orc_heap_item *heap = (orc_heap_item *)malloc(cap * sizeof(orc_heap_item));
    if (!heap)
        return -2;
    heap[len++] = (orc_heap_item){0.0f, (uint32_t)start_node};
    while (len) {
        const orc_heap_item top = heap[0];
        heap[0] = heap[--len];
        for (size_t i = 0;;) { /* sift down */
            size_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < len && heap[l].d < heap[m].d)
                m = l;
            if (r < len && heap[r].d < heap[m].d)
                m = r;
            if (m == i)
                break;
            const orc_heap_item t = heap[i];
            heap[i] = heap[m];
            heap[m] = t;
            i = m;
        }
        if (top.d > dist[top.v])
            continue;
        for (uint32_t e = off[top.v]; e < off[top.v + 1]; ++e) {
            const float nd = top.d + w[e];
            if (nd < dist[tgt[e]]) {
                dist[tgt[e]] = nd;
                if (len == cap) {
                    cap *= 2;
                    heap = (orc_heap_item *)realloc(heap, cap * sizeof(orc_heap_item));
                    if (!heap)
                        return -2;
                }
                size_t i = len++;
                heap[i] = (orc_heap_item){nd, tgt[e]};
                while (i && heap[(i - 1) / 2].d > heap[i].d) { /* sift up */
                    const orc_heap_item t = heap[i];
                    heap[i] = heap[(i - 1) / 2];
                    heap[(i - 1) / 2] = t;
                    i = (i - 1) / 2;
                }
            }
        }
    }
    free(heap);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Global triangle count — crates/algos/src/triangle_count.rs:47-70 with the put-back iterator
 * of crates/algos/src/utils.rs:8-101: for u; for v in N(u) while v <= u; cursor over N(u)
 * restarted for every v; for w in N(v) while w <= v: advance cursor while x < w; at the first
 * x >= w count if equal and do NOT consume x.  Lists must be sorted.  Threaded over 64-node
 * chunks like the reference (:10, :36-45); threads == 1 gives the sequential order.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n;
    const uint32_t *off, *tgt;
    atomic_uint_fast64_t *next_chunk;
    uint64_t count;
} orc_tc_job;

static void *orc_tc_worker(void *arg)
{
    orc_tc_job *j = (orc_tc_job *)arg;
    const uint32_t *off = j->off, *tgt = j->tgt;
    uint64_t tri = 0;
    for (;;) {
        uint64_t start = atomic_fetch_add(j->next_chunk, 64);
        if (start >= j->n)
            break;
        uint64_t end = start + 64 > j->n ? j->n : start + 64;
        for (uint32_t u = (uint32_t)start; u < (uint32_t)end; ++u) {
            const uint32_t us = off[u], ue = off[u + 1];
            for (uint32_t a = us; a < ue; ++a) {
                uint32_t v = tgt[a];
                if (v > u)
                    break;
                uint32_t it = us;
                for (uint32_t b = off[v]; b < off[v + 1]; ++b) {
                    uint32_t w = tgt[b];
                    if (w > v)
                        break;
                    while (it < ue) {
                        uint32_t x = tgt[it];
                        if (x >= w) {
                            if (x == w)
                                tri += 1;
                            break; /* x is put back: cursor does not advance */
                        }
                        ++it;
                    }
                }
            }
        }
    }
    j->count = tri;
    return NULL;
}

ORC_EXPORT uint64_t orc_triangle_count(uint32_t n, const uint32_t *off, const uint32_t *tgt, uint32_t threads)
{
    if (threads == 0)
        threads = 1;
    atomic_uint_fast64_t next;
    atomic_init(&next, 0);
    pthread_t *tid = (pthread_t *)malloc(threads * sizeof(pthread_t));
    orc_tc_job *jobs = (orc_tc_job *)malloc(threads * sizeof(orc_tc_job));
    uint64_t total = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        jobs[t] = (orc_tc_job){n, off, tgt, &next, 0};
        pthread_create(&tid[t], NULL, orc_tc_worker, &jobs[t]);
    }
    for (uint32_t t = 0; t < threads; ++t) {
        pthread_join(tid[t], NULL);
        total += jobs[t].count;
    }
    free(tid);
    free(jobs);
    return total;
}

/* Greedy degree partition — crates/builder/src/graph_ops.rs:431-439 (in_degree_partition:
 * batch = ceil(edge_count / concurrency)) and :479-509 (greedy_node_map_partition): walk nodes
 * in order, close a range once its accumulated degree reaches the batch size — but only while
 * fewer than concurrency-1 ranges exist; the last range absorbs the rest.  ranges_out: 2*concurrency u32
 * (start,end pairs); returns the number of ranges produced. */
ORC_EXPORT uint32_t orc_greedy_degree_partition(uint32_t n, const uint32_t *off, uint32_t concurrency,
                                                uint32_t *ranges_out)
{
    uint64_t total = off[n];
    uint64_t batch = (total + concurrency - 1) / concurrency;
    uint32_t count = 0, start = 0;
    uint64_t acc = 0;
    for (uint32_t u = 0; u < n; ++u) {
        acc += off[u + 1] - off[u];
        if ((count < concurrency - 1 && acc >= batch) || u == n - 1) {
            ranges_out[2 * count] = start;
            ranges_out[2 * count + 1] = u + 1;
            ++count;
            start = u + 1;
            acc = 0;
        }
    }
    return count;
}
