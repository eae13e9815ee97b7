// device_utils.hpp — wavefront-64 helpers for gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gm {

constexpr int kWave = 64; // CDNA wavefront width (hard-coded: gfx950 only)

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        uint32_t w = __shfl_xor(v, o, kWave);
        v = w < v ? w : v;
    }
    return v;
}

// Deterministic block-wide sum for blockDim.x == 64 * NWAVES; result valid in thread 0.
template <class T, int NWAVES> __device__ __forceinline__ T block_sum(T v, T *lds /* NWAVES */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    if (lane == 0)
        lds[wave] = v;
    __syncthreads();
    T total = T(0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w)
            total += lds[w];
    }
    return total;
}

// L1-bypassing (agent-scope, relaxed) accesses for data other workgroups mutate inside the same
// launch — lowers to global_load/store ... sc1 on gfx950 (MI355X_MICROARCH.md, inter-workgroup
// visibility table): a plain load may keep returning a stale L1 line forever.
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// first index i in [lo, hi) with key(i) >= target, or hi
template <class KeyFn> __device__ __forceinline__ uint64_t lower_bound_fn(uint64_t lo, uint64_t hi, uint64_t target, KeyFn key)
{
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (key(mid) < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// row of CSR entry e: largest r with off[r] <= e (skips empty rows)   [off has n+1 entries]
__device__ __forceinline__ uint32_t row_of_entry(const uint32_t *__restrict__ off, uint32_t n, uint32_t e)
{
    uint32_t lo = 0, hi = n; // invariant: off[lo] <= e < off[hi]
    while (hi - lo > 1) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (off[mid] <= e)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

} // namespace gm
