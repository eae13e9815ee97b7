// fetchcal.hip — what rocprofv3's FETCH_SIZE (and the L2 miss count) report for the ACCESS PATTERNS the three `extra`
// algorithms are made of, each with a byte count known in advance (the guide calibrates FETCH_SIZE for wide
// streaming reads only — "exactly half of the bytes of a 16 B/lane streaming read on gfx950; other access widths uncalibrated",
// MI355X_MICROARCH.md — while the triangle count's "fraction by counters" applies that factor 2 to random 128-byte records and
// 2-byte id streams, and SSSP / WCC apply it to random 4-byte probes).
//
//   cal_stream16   1 GiB streamed, 16 B per lane                                   (the guide's own case: expect bytes / 2)
//   cal_stream8    1 GiB streamed, 8 B per lane: four 2-byte ids                   (tc_rows_kernel's long fronts, pb p2_dst)
//   cal_rec128     64 M random 128-byte records out of 2 GiB, 8 lanes x 16 B each  (tc_rows_kernel's list records)
//   cal_gather4    256 M random 4-byte loads out of 2 GiB, 8 in flight per lane    (SSSP's distance probes, WCC's parents)
//   cal_gather4sc1 the same through agent-scope (sc1) loads                        (ld_agent: what those kernels issue)
//   cal_write4     256 M random 4-byte stores into 2 GiB                           (WRITE_SIZE for scattered stores)
//
// Every table is far larger than the 256 MiB Infinity Cache + the L2s, so nearly every access misses them; the random
// indices are computed in registers (no index stream).  Run under `rocprofv3 --pmc <counter> --kernel-trace`; the kernels'
// names carry the pattern; tools/fetchcal_report.py divides counters by the known bytes.  Prints its own timings as well.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                                               \
    do {                                                                                                                    \
        hipError_t e_ = (x);                                                                                                \
        if (e_ != hipSuccess) {                                                                                             \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);                                           \
            return 1;                                                                                                       \
        }                                                                                                                   \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void cal_stream16(const uint4 *__restrict__ in, uint64_t n16, uint32_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        *out = acc;
}

__global__ __launch_bounds__(256) void cal_stream8(const uint2 *__restrict__ in, uint64_t n8, uint32_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint2 v = in[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u)
        *out = acc;
}

// record r of `records` (a power of two) per 8-lane group and step, 4 steps in flight
__global__ __launch_bounds__(256) void cal_rec128(const uint4 *__restrict__ table, uint64_t records, uint64_t reads, uint32_t *out)
{
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8, groups = (uint64_t)gridDim.x * blockDim.x / 8;
    const uint32_t l = threadIdx.x & 7u;
    uint32_t acc = 0;
    for (uint64_t k = group; k < reads; k += groups * 4) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t kk = k + (uint64_t)j * groups;
            const uint64_t r = mix(kk) & (records - 1);
            v[j] = kk < reads ? table[r * 8 + l] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u)
        *out = acc;
}

template <bool SC1>
__global__ __launch_bounds__(256) void cal_gather4(const uint32_t *__restrict__ table, uint64_t words, uint64_t reads, uint32_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, threads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t k = t; k < reads; k += threads * 8) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t kk = k + (uint64_t)j * threads;
            const uint64_t w = mix(kk) & (words - 1);
            if (kk < reads)
                v[j] = SC1 ? __hip_atomic_load(table + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : table[w];
            else
                v[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc ^= v[j];
    }
    if (acc == 0x12345678u)
        *out = acc;
}

__global__ __launch_bounds__(256) void cal_write4(uint32_t *__restrict__ table, uint64_t words, uint64_t writes)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, threads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = t; k < writes; k += threads)
        table[mix(k) & (words - 1)] = (uint32_t)k;
}

int main()
{
    const uint64_t table_bytes = 2ull << 30, stream_bytes = 1ull << 30;
    const uint64_t rec_reads = 64ull << 20, gathers = 256ull << 20;
    uint32_t *table, *out;
    CK(hipMalloc(&table, table_bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(table, 1, table_bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms = 0;
#define TIMED(name, bytes, ...)                                                                                             \
    for (int rep = 0; rep < 2; ++rep) {                                                                                     \
        CK(hipEventRecord(a));                                                                                              \
        __VA_ARGS__;                                                                                                        \
        CK(hipEventRecord(b));                                                                                              \
        CK(hipEventSynchronize(b));                                                                                         \
        CK(hipEventElapsedTime(&ms, a, b));                                                                                 \
    }                                                                                                                       \
    printf("%-16s requested bytes %llu  %.3f ms  %.1f GB/s\n", name, (unsigned long long)(bytes), ms, (double)(bytes) / ms / 1e6)
    TIMED("cal_stream16", stream_bytes, hipLaunchKernelGGL(cal_stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)table, stream_bytes / 16, out));
    TIMED("cal_stream8", stream_bytes, hipLaunchKernelGGL(cal_stream8, dim3(256 * 16), dim3(256), 0, 0, (const uint2 *)table, stream_bytes / 8, out));
    TIMED("cal_rec128", rec_reads * 128, hipLaunchKernelGGL(cal_rec128, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)table, table_bytes / 128, rec_reads, out));
    TIMED("cal_gather4", gathers * 4, hipLaunchKernelGGL(cal_gather4<false>, dim3(256 * 16), dim3(256), 0, 0, table, table_bytes / 4, gathers, out));
    TIMED("cal_gather4sc1", gathers * 4, hipLaunchKernelGGL(cal_gather4<true>, dim3(256 * 16), dim3(256), 0, 0, table, table_bytes / 4, gathers, out));
    TIMED("cal_write4", gathers * 4, hipLaunchKernelGGL(cal_write4, dim3(256 * 16), dim3(256), 0, 0, table, table_bytes / 4, gathers));
    return 0;
}
