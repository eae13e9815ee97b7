// streams_repro.hip — torch-free reproducer of the stream-per-part sweep schedule of graph_amd/distributed.py
// (PiecewiseExchange._sweep_streams, removed in round 6) + the engine's fork / join of the hub kernels
// (pagerank_pb.hip: pb_sweep_accum_part), as N PROCESSES ON ONE GPU.  Round 5 saw that schedule compute wrong intermediate
// sweeps with 8 gloo processes on one device (profiles/r05_multi_rank_streams.txt): a consumer on one stream ran before its
// producer on another although an event orders them.  Here every kernel only STAMPS the sweep number into what it
// produces and CHECKS the stamp of what it consumes, so that a dependency that did not hold is counted where it happens:
//
//   per sweep s (x[cur] = the exchanged vector, region k = half k; vals = the value stream; out = this rank's rows):
//     stream k:  wait `start`, wait ev_acc[*] of s-1, wait the exchange of region k     -> bin_k   : x[cur] region k == s  -> vals (region k's share) = s + 1
//     main:      wait ev_bin[*]                                                         -> hot_k   : x[cur] == s           -> hot = s + 1
//     stream k:  wait ev_bin[*], ev_hot
//        part 0: ev_fork(main) -> side [-> chain: long_k; side: seq_k; join chain] -> ev_join(side)
//                long_k / seq_k: vals (hub share) == s + 1, hot == s + 1 -> out[hub rows] = s + 1        (low-priority streams, slow small grids)
//                accum_k(part k): vals (part k's bins) == s + 1, hot == s + 1 -> out[part k's ordinary rows] = s + 1
//                wait ev_join
//                compact_k: out[part k rows, hub rows included] == s + 1 -> send[k] = s + 1
//                exchange of region k: copy stream waits an event of stream k, D2H, a host thread checks, meets the other
//                processes at a barrier in shared memory, H2D into x[1 - cur] region k, event        (what gloo does)
//                ev_acc[k]
//     main:      wait ev_acc[*] -> err_k: out == s + 1 everywhere
//
// Usage: tools/streams_repro [--procs 8] [--sweeps 200] [--streams 1] [--null-main 1] [--fresh-events 1] [--side-prio 1]
//                            [--fork 1] [--thread 1] [--lockstep 1] [--mb 64]
// Prints one line per process: violations per check, the first one with its sweep.  Exit code 1 if any.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <string>
#include <sys/mman.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>

#define CK(x)                                                                                                               \
    do {                                                                                                                    \
        hipError_t e_ = (x);                                                                                                \
        if (e_ != hipSuccess) {                                                                                             \
            fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);                                  \
            _exit(3);                                                                                                       \
        }                                                                                                                   \
    } while (0)

enum Check { C_BIN_X = 0, C_HOT_X, C_ACC_VALS, C_ACC_HOT, C_HUB_VALS, C_HUB_HOT, C_COMPACT, C_COMPACT_HUB, C_ERR, C_ERR_HUB, C_HOST, C_COUNT };
static const char *check_names[C_COUNT] = {"bin:x", "hot:x", "accum:vals", "accum:hot", "hub:vals", "hub:hot", "compact:rows",
                                           "compact:hubrows", "err:rows", "err:hubrows", "host:send"};

struct Report {
    unsigned long long bad[C_COUNT];
    unsigned long long first_sweep[C_COUNT]; // sweep of the first violation + 1 (0: none)
    unsigned long long first_got[C_COUNT];
};

__device__ void report(Report *r, int c, uint32_t sweep, uint32_t got)
{
    if (atomicAdd(&r->bad[c], 1ull) == 0ull) {
        r->first_sweep[c] = sweep + 1ull;
        r->first_got[c] = got;
    }
}

constexpr uint32_t HUB_EVERY = 1024; // every 1024th row is a "hub row", written by the hub kernels
constexpr uint32_t CHUNK = 4096;     // vals: chunk c belongs to source region c % parts

// x[cur] region -> its share of vals
__global__ void bin_k(const uint32_t *x, uint32_t x_lo, uint32_t x_hi, uint32_t *vals, uint32_t nv, uint32_t region, uint32_t parts,
                      uint32_t sweep, Report *rep)
{
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = x_lo + t; i < x_hi; i += stride)
        if (x[i] != sweep)
            report(rep, C_BIN_X, sweep, x[i]);
    for (uint32_t i = t; i < nv; i += stride)
        if ((i / CHUNK) % parts == region)
            vals[i] = sweep + 1u;
}

__global__ void hot_k(const uint32_t *x, uint32_t n, uint32_t *hot, uint32_t nh, uint32_t sweep, Report *rep)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nh) {
        const uint32_t v = x[(uint32_t)(((uint64_t)t * 2654435761ull) % n)];
        if (v != sweep)
            report(rep, C_HOT_X, sweep, v);
        hot[t] = sweep + 1u;
    }
}

// vals [v_lo, v_hi) (every region's share of it) + hot -> out rows [r_lo, r_hi) that are not hub rows
__global__ void accum_k(const uint32_t *vals, uint32_t v_lo, uint32_t v_hi, const uint32_t *hot, uint32_t nh, uint32_t *out,
                        uint32_t r_lo, uint32_t r_hi, uint32_t sweep, Report *rep)
{
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = v_lo + t; i < v_hi; i += stride)
        if (vals[i] != sweep + 1u)
            report(rep, C_ACC_VALS, sweep, vals[i]);
    for (uint32_t i = t; i < nh; i += stride)
        if (hot[i] != sweep + 1u)
            report(rep, C_ACC_HOT, sweep, hot[i]);
    for (uint32_t r = r_lo + t; r < r_hi; r += stride)
        if (r % HUB_EVERY != 0)
            out[r] = sweep + 1u;
}

// a slow, small grid: every thread walks a chain of dependent loads through its hub share of vals, then finishes hub rows
// `which` of every pair (0: the lane walks' rows, 1: the long rows')
__global__ void hub_k(const uint32_t *vals, uint32_t v_lo, uint32_t v_hi, const uint32_t *hot, uint32_t nh, uint32_t *out, uint32_t n,
                      uint32_t which, uint32_t chain, uint32_t sweep, Report *rep)
{
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t at = v_lo + t % (v_hi - v_lo);
    for (uint32_t k = 0; k < chain; ++k) { // dependent: the next address comes from the value read
        const uint32_t v = __builtin_nontemporal_load(vals + at);
        if (v != sweep + 1u)
            report(rep, C_HUB_VALS, sweep, v);
        at = v_lo + (uint32_t)(((uint64_t)at * 1664525ull + v + 1013904223ull) % (v_hi - v_lo));
    }
    if (t < nh && hot[t] != sweep + 1u)
        report(rep, C_HUB_HOT, sweep, hot[t]);
    const uint32_t hubs = (n + HUB_EVERY - 1u) / HUB_EVERY;
    for (uint32_t h = stride - 1u - t; h < hubs; h += stride) // (the LAST workgroups finish the rows: the tail of the kernel)
        if ((h & 1u) == which)
            out[h * HUB_EVERY] = sweep + 1u + (at == 0xFFFFFFFFu ? 1u : 0u);
}

__global__ void compact_k(const uint32_t *out, uint32_t r_lo, uint32_t r_hi, uint32_t *send, uint32_t sweep, Report *rep)
{
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t r = r_lo + t; r < r_hi; r += stride) {
        const uint32_t v = out[r];
        if (v != sweep + 1u)
            report(rep, r % HUB_EVERY ? C_COMPACT : C_COMPACT_HUB, sweep, v);
        send[r - r_lo] = v;
    }
}

__global__ void err_k(const uint32_t *out, uint32_t n, uint32_t sweep, Report *rep)
{
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t r = t; r < n; r += stride)
        if (out[r] != sweep + 1u)
            report(rep, r % HUB_EVERY ? C_ERR : C_ERR_HUB, sweep, out[r]);
}

__global__ void fill_k(uint32_t *p, uint32_t n, uint32_t v)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        p[i] = v;
}

struct Shared { // one page shared by the processes: the barrier the "collective" meets at
    std::atomic<uint32_t> arrived[4096];
};

struct Opt {
    int procs = 8, sweeps = 200, streams = 1, null_main = 1, fresh_events = 1, side_prio = 1, fork_hub = 1, thread = 1, lockstep = 1, mb = 64;
    int chain = 64, parts = 2;
};

static int child(int rank, const Opt &o, Shared *sh)
{
    CK(hipSetDevice(0));
    const uint32_t parts = (uint32_t)o.parts;
    const uint32_t nv = (uint32_t)o.mb * (1u << 18); // vals entries (u32)
    const uint32_t n = 1u << 21, nh = 1u << 14;      // rows, hot entries
    uint32_t *vals, *hot, *out, *x[2], *send[8];
    Report *rep;
    CK(hipMalloc(&vals, (size_t)nv * 4));
    CK(hipMalloc(&hot, nh * 4));
    CK(hipMalloc(&out, n * 4));
    CK(hipMalloc(&x[0], n * 4));
    CK(hipMalloc(&x[1], n * 4));
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMemset(rep, 0, sizeof(Report)));
    const uint32_t half = n / parts;
    uint32_t *pin_in[8], *pin_out[8];
    for (uint32_t k = 0; k < parts; ++k) {
        CK(hipMalloc(&send[k], half * 4));
        CK(hipHostMalloc(&pin_in[k], half * 4));
        CK(hipHostMalloc(&pin_out[k], half * 4));
    }
    int least = 0, greatest = 0;
    if (o.side_prio)
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t strs[8] = {nullptr}, side, chain, copy_in[8], copy_out[8];
    if (!o.null_main)
        CK(hipStreamCreateWithFlags(&strs[0], hipStreamNonBlocking));
    for (uint32_t k = 1; k < parts; ++k)
        CK(hipStreamCreateWithFlags(&strs[k], hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&chain, hipStreamNonBlocking, least));
    for (uint32_t k = 0; k < parts; ++k) {
        CK(hipStreamCreateWithFlags(&copy_in[k], hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&copy_out[k], hipStreamNonBlocking));
    }
    hipEvent_t ev_fork, ev_join, ev_cfork, ev_cjoin;
    CK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_cfork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_cjoin, hipEventDisableTiming));
    // events of the schedule: fresh ones every sweep, destroyed at the next (what torch.cuda.Event objects do), or reused
    auto new_event = [&]() {
        hipEvent_t e;
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    };
    std::vector<hipEvent_t> garbage;
    auto drop = [&](hipEvent_t e) {
        if (o.fresh_events)
            garbage.push_back(e);
    };
    fill_k<<<256, 256, 0, strs[0]>>>(x[0], n, 0u);
    fill_k<<<256, 256, 0, strs[0]>>>(x[1], n, 0xDEADu);
    fill_k<<<256, 256, 0, strs[0]>>>(out, n, 0xBEEFu);
    CK(hipDeviceSynchronize());

    struct Work {
        std::future<void> done;
        hipEvent_t ev_out = nullptr;
        bool active = false;
    } works[8];
    unsigned long long host_bad = 0, host_first = 0;
    uint32_t barrier_no = 0;
    // the exchange of region k, issued from stream `st` (gloo: ProcessGroupGloo's AsyncAllgatherCUDAWork)
    auto start_gather = [&](uint32_t buf, uint32_t k, hipStream_t st, uint32_t sweep) {
        compact_k<<<128, 256, 0, st>>>(out, k * half, (k + 1) * half, send[k], sweep, rep);
        hipEvent_t e_in = new_event();
        CK(hipEventRecord(e_in, st)); // inputs and outputs: the copy streams wait for what the issuing stream has enqueued
        CK(hipStreamWaitEvent(copy_in[k], e_in, 0));
        CK(hipStreamWaitEvent(copy_out[k], e_in, 0));
        CK(hipMemcpyAsync(pin_in[k], send[k], half * 4, hipMemcpyDeviceToHost, copy_in[k]));
        hipEvent_t ev_out = new_event();
        const uint32_t my_barrier = barrier_no++;
        auto body = [&, buf, k, sweep, ev_out, my_barrier]() {
            CK(hipSetDevice(0));
            CK(hipStreamSynchronize(copy_in[k]));
            CK(hipStreamSynchronize(copy_out[k]));
            for (uint32_t i = 0; i < half; i += 97)
                if (pin_in[k][i] != sweep + 1u) {
                    if (!host_bad++)
                        host_first = sweep + 1ull;
                }
            if (o.lockstep) { // every process's exchange number `my_barrier` has arrived
                std::atomic<uint32_t> &slot = sh->arrived[my_barrier % 4096];
                const uint32_t want = (uint32_t)o.procs * (my_barrier / 4096 + 1);
                slot.fetch_add(1);
                for (uint64_t spins = 0; slot.load() < want; ++spins) {
                    std::this_thread::yield();
                    if (spins > 400000000ull) { // a process died: do not hang the box
                        fprintf(stderr, "barrier %u timed out\n", my_barrier);
                        _exit(4);
                    }
                }
            }
            memcpy(pin_out[k], pin_in[k], half * 4);
            CK(hipMemcpyAsync(x[buf] + k * half, pin_out[k], half * 4, hipMemcpyHostToDevice, copy_out[k]));
            CK(hipEventRecord(ev_out, copy_out[k]));
        };
        works[k].ev_out = ev_out;
        works[k].active = true;
        if (o.thread)
            works[k].done = std::async(std::launch::async, body);
        else
            body();
        drop(e_in);
    };
    auto wait_work = [&](uint32_t k, hipStream_t st) { // work.wait(): the host blocks until the copy back is issued, the stream waits for it
        if (!works[k].active)
            return;
        if (o.thread)
            works[k].done.get();
        CK(hipStreamWaitEvent(st, works[k].ev_out, 0));
        drop(works[k].ev_out);
        works[k].active = false;
    };
    const uint32_t hub_lo = nv - nv / 8, vper = (nv - nv / 8) / parts;
    auto hub_dispatch = [&](hipStream_t st, uint32_t sweep) { // pb_hub_dispatch: the long rows on `chain` beside the lane walks
        CK(hipEventRecord(ev_cfork, st));
        CK(hipStreamWaitEvent(chain, ev_cfork, 0));
        hub_k<<<48, 512, 0, chain>>>(vals, hub_lo, nv, hot, nh, out, n, 1u, (uint32_t)o.chain, sweep, rep);
        CK(hipEventRecord(ev_cjoin, chain));
        hub_k<<<96, 256, 0, st>>>(vals, hub_lo, nv, hot, nh, out, n, 0u, (uint32_t)o.chain * 2u, sweep, rep);
        CK(hipStreamWaitEvent(st, ev_cjoin, 0));
    };
    auto accum_part = [&](uint32_t k, hipStream_t st, uint32_t sweep) { // pb_sweep_accum_part
        if (k == 0) {
            if (o.fork_hub) {
                CK(hipEventRecord(ev_fork, st));
                CK(hipStreamWaitEvent(side, ev_fork, 0));
                hub_dispatch(side, sweep);
                CK(hipEventRecord(ev_join, side));
            } else {
                hub_dispatch(st, sweep);
                CK(hipEventRecord(ev_join, st));
            }
        }
        accum_k<<<512, 256, 0, st>>>(vals, k * vper, (k + 1) * vper, hot, nh, out, k * half, (k + 1) * half, sweep, rep);
        if (o.fork_hub || k != 0)
            CK(hipStreamWaitEvent(st, ev_join, 0));
    };

    uint32_t cur = 0;
    std::vector<hipEvent_t> ev_acc_prev;
    for (uint32_t k = 0; k < parts; ++k) // ex.start(): x[0] holds stamp 0 already; nothing in flight
        works[k].active = false;
    for (uint32_t s = 0; s < (uint32_t)o.sweeps; ++s) {
        if (o.streams) {
            hipEvent_t start = new_event();
            CK(hipEventRecord(start, strs[0]));
            std::vector<hipEvent_t> ev_bin, ev_acc;
            for (uint32_t k = 0; k < parts; ++k) {
                CK(hipStreamWaitEvent(strs[k], start, 0));
                for (hipEvent_t e : ev_acc_prev)
                    CK(hipStreamWaitEvent(strs[k], e, 0));
                wait_work(k, strs[k]);
                bin_k<<<512, 256, 0, strs[k]>>>(x[cur], k * half, (k + 1) * half, vals, nv, k, parts, s, rep);
                hipEvent_t e = new_event();
                CK(hipEventRecord(e, strs[k]));
                ev_bin.push_back(e);
            }
            for (hipEvent_t e : ev_bin)
                CK(hipStreamWaitEvent(strs[0], e, 0));
            hot_k<<<(nh + 255) / 256, 256, 0, strs[0]>>>(x[cur], n, hot, nh, s, rep);
            hipEvent_t ev_hot = new_event();
            CK(hipEventRecord(ev_hot, strs[0]));
            for (uint32_t k = 0; k < parts; ++k) {
                for (hipEvent_t e : ev_bin)
                    CK(hipStreamWaitEvent(strs[k], e, 0));
                CK(hipStreamWaitEvent(strs[k], ev_hot, 0));
                accum_part(k, strs[k], s);
                start_gather(1 - cur, k, strs[k], s);
                hipEvent_t e = new_event();
                CK(hipEventRecord(e, strs[k]));
                ev_acc.push_back(e);
            }
            for (hipEvent_t e : ev_acc)
                CK(hipStreamWaitEvent(strs[0], e, 0));
            for (hipEvent_t e : ev_acc_prev)
                drop(e);
            ev_acc_prev = ev_acc;
            drop(start), drop(ev_hot);
            for (hipEvent_t e : ev_bin)
                drop(e);
        } else { // the in-order schedule: every part on the caller's stream
            for (uint32_t k = 0; k < parts; ++k) {
                wait_work(k, strs[0]);
                bin_k<<<512, 256, 0, strs[0]>>>(x[cur], k * half, (k + 1) * half, vals, nv, k, parts, s, rep);
            }
            for (uint32_t k = 0; k < parts; ++k) {
                if (k == 0)
                    hot_k<<<(nh + 255) / 256, 256, 0, strs[0]>>>(x[cur], n, hot, nh, s, rep);
                accum_part(k, strs[0], s);
                start_gather(1 - cur, k, strs[0], s);
            }
        }
        err_k<<<256, 256, 0, strs[0]>>>(out, n, s, rep);
        cur = 1 - cur;
        // what Python's garbage collection does to the per-sweep events: destroyed while waits on them may be pending
        if (garbage.size() > 64) {
            for (size_t i = 0; i + 32 < garbage.size(); ++i)
                CK(hipEventDestroy(garbage[i]));
            garbage.erase(garbage.begin(), garbage.end() - 32);
        }
    }
    for (uint32_t k = 0; k < parts; ++k)
        wait_work(k, strs[0]);
    CK(hipDeviceSynchronize());
    Report r;
    CK(hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost));
    r.bad[C_HOST] = host_bad, r.first_sweep[C_HOST] = host_first;
    unsigned long long total = 0;
    std::string line = "rank " + std::to_string(rank) + ":";
    for (int c = 0; c < C_COUNT; ++c) {
        total += r.bad[c];
        if (r.bad[c])
            line += " " + std::string(check_names[c]) + "=" + std::to_string(r.bad[c]) + "(first at sweep " +
                    std::to_string(r.first_sweep[c] - 1) + ", saw stamp " + std::to_string(r.first_got[c]) + ")";
    }
    if (!total)
        line += " every dependency held over " + std::to_string(o.sweeps) + " sweeps";
    printf("%s\n", line.c_str());
    fflush(stdout);
    return total ? 1 : 0;
}

int main(int argc, char **argv)
{
    Opt o;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const int v = atoi(argv[i + 1]);
        if (a == "--procs") o.procs = v;
        else if (a == "--sweeps") o.sweeps = v;
        else if (a == "--streams") o.streams = v;
        else if (a == "--null-main") o.null_main = v;
        else if (a == "--fresh-events") o.fresh_events = v;
        else if (a == "--side-prio") o.side_prio = v;
        else if (a == "--fork") o.fork_hub = v;
        else if (a == "--thread") o.thread = v;
        else if (a == "--lockstep") o.lockstep = v;
        else if (a == "--mb") o.mb = v;
        else if (a == "--chain") o.chain = v;
        else if (a == "--parts") o.parts = v;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.parts < 1 || o.parts > 8 || o.procs < 1) return 2;
    Shared *sh = (Shared *)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) return 2;
    memset((void *)sh, 0, sizeof(Shared));
    std::vector<pid_t> kids;
    for (int r = 0; r < o.procs; ++r) { // fork BEFORE the first HIP call: every process gets a runtime of its own
        const pid_t p = fork();
        if (p == 0)
            _exit(child(r, o, sh));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
    }
    printf("streams_repro procs=%d sweeps=%d streams=%d null_main=%d fresh_events=%d side_prio=%d fork=%d thread=%d lockstep=%d mb=%d chain=%d: %d of %d processes saw a violated dependency\n",
           o.procs, o.sweeps, o.streams, o.null_main, o.fresh_events, o.side_prio, o.fork_hub, o.thread, o.lockstep, o.mb, o.chain, bad, o.procs);
    return bad ? 1 : 0;
}
