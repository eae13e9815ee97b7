// membench.hip — memory-system ceilings that bound the PageRank pull sweep on MI355X:
// streaming read rate, and random 4-byte gather rate as a function of the gathered table size
// (L2-resident / Infinity-Cache-resident / HBM-resident), with 8 independent gathers per lane.
// Usage: tools/membench   (prints one line per experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stream_read(const uint4 *__restrict__ in, uint64_t n16, uint32_t *out)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}

// float4 copy (the guide's 6.29 TB/s figure counts read + written bytes)
__global__ __launch_bounds__(256) void stream_copy(const float4 *__restrict__ in, float4 *__restrict__ out, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        out[i] = in[i];
}

// the byte mix of pb_bin_kernel: read 2 B/entry (coalesced), write 4 B/entry in runs of `run` entries
// (run*4 bytes, 16-byte aligned) whose destinations are scattered (a multiplicative permutation of
// the run index), every wavefront writing whole 1 KiB pieces of a run.  run = 0: contiguous.
__global__ __launch_bounds__(1024) void bin_like(const uint2 *__restrict__ in, float4 *__restrict__ out, uint64_t n4, uint32_t run4,
                                                 uint64_t nruns, uint64_t mult)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint2 v = in[i];
        uint64_t o = i;
        if (run4) {
            const uint64_t r = i / run4, k = i % run4;
            o = ((r * mult) % nruns) * run4 + k;
        }
        out[o] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.x >> 16), __uint_as_float(v.y >> 16));
    }
}

// idx: m random indices (streamed, coalesced); table: gathered; 8 per lane in flight
__global__ __launch_bounds__(256) void gather8(const uint32_t *__restrict__ idx, const float *__restrict__ table,
                                               uint64_t m, float *out)
{
    uint64_t base = (uint64_t)blockIdx.x * 2048;
    uint64_t stride = (uint64_t)gridDim.x * 2048;
    float acc = 0.f;
    for (; base < m; base += stride) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { uint64_t i = base + threadIdx.x + k * 256; v[k] = i < m ? idx[i] : 0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += table[v[k]];
    }
    if (acc == 1.2345f) *out = acc;
}

// LDS atomic throughput: every lane adds into a pseudo-random slot of an LDS table
template <class T, int SLOTS>
__global__ __launch_bounds__(1024) void lds_atomic(int iters, T *out)
{
    __shared__ T tab[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += 1024) tab[i] = T(0);
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    for (int i = 0; i < iters; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        atomicAdd(&tab[x % SLOTS], T(1));
    }
    __syncthreads();
    if (threadIdx.x == 0 && tab[0] == T(123456789)) *out = tab[1];
}

template <class T, int SLOTS> static void run_lds(const char *name, T *out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    float best = 1e30f, ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((lds_atomic<T, SLOTS>), dim3(256 * 4), dim3(1024), 0, 0, iters, out);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("lds_atomic %-28s: %.3f ms  %.1f Gatomic/s\n", name, best, 256.0 * 4 * 1024 * iters / best / 1e6);
}

static uint64_t sm64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ void fill_idx(uint32_t *idx, uint64_t m, uint32_t mask, int skew)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        uint64_t x = i * 0xD1342543DE82EF95ull + 12345;
        x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        uint32_t r = (uint32_t)x & mask;
        if (skew) { // product of two uniforms -> skewed toward small ids
            uint32_t r2 = (uint32_t)(x >> 32) & mask;
            r = (uint32_t)(((uint64_t)r * r2) >> __builtin_ctz(mask + 1));
        }
        idx[i] = r;
    }
}

int main()
{
    const uint64_t m = 1ull << 28; // 268M indices = 1 GiB
    uint32_t *idx; float *table, *out;
    CK(hipMalloc(&idx, m * 4));
    CK(hipMalloc(&table, (1ull << 28) * 4));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 0, (1ull << 28) * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    // streaming read
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)table, (1ull << 28) * 4 / 16, (uint32_t *)out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        printf("stream_read 1GiB: %.3f ms  %.1f GB/s\n", ms, (1ull << 30) / ms / 1e6);
    }
    {   // copy + bin-like write mixes
        float *dst;
        CK(hipMalloc(&dst, 1ull << 30));
        CK(hipMemset(dst, 0, 1ull << 30));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(stream_copy, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)table, (float4 *)dst, (1ull << 30) / 16);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            printf("stream_copy 1GiB->1GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * (1ull << 30) / ms / 1e6);
        }
        const uint64_t n4 = (1ull << 30) / 16; // 64M float4 = 256M entries
        const uint32_t runs4[] = {0, 8, 16, 28, 32, 56, 64, 128, 256, 1024};
        for (uint32_t run4 : runs4) {
            const uint64_t nruns = run4 ? n4 / run4 : 1;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(bin_like, dim3(512), dim3(1024), 0, 0, (const uint2 *)table, (float4 *)dst, n4, run4, nruns,
                                   (uint64_t)1000003);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            printf("bin_like run=%5u B: %.3f ms  %.1f GB/s (2 B read + 4 B written per entry)\n", run4 * 16, ms,
                   (double)n4 * 24 / ms / 1e6);
        }
        CK(hipFree(dst));
    }
    run_lds<unsigned long long, 16384>("u64 x 16384 slots (128KiB)", (unsigned long long *)out);
    run_lds<unsigned long long, 2048>("u64 x 2048 slots", (unsigned long long *)out);
    run_lds<unsigned int, 16384>("u32 x 16384 slots", (unsigned int *)out);
    run_lds<float, 16384>("f32 x 16384 slots", (float *)out);
    run_lds<unsigned long long, 64>("u64 x 64 slots (conflicts)", (unsigned long long *)out);
    run_lds<unsigned long long, 1>("u64 x 1 slot (all collide)", (unsigned long long *)out);
    if (getenv("MEMBENCH_LDS_ONLY")) return 0;
    for (int skew = 0; skew < 2; ++skew)
        for (int lg = 18; lg <= 28; lg += 2) {
            uint32_t mask = (1u << lg) - 1;
            hipLaunchKernelGGL(fill_idx, dim3(8192), dim3(256), 0, 0, idx, m, mask, skew);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(gather8, dim3(256 * 8), dim3(256), 0, 0, idx, table, m, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            printf("gather8 %s table=%4u MiB: %.3f ms  %.2f Ggather/s  (idx stream %.1f GB/s)\n", skew ? "skewed " : "uniform",
                   (unsigned)((4ull << lg) >> 20), best, m / best / 1e6, m * 4 / best / 1e6);
        }
    return 0;
}
