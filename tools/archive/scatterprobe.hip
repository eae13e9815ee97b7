// scatterprobe.hip — which stretches of physical memory go together for SCATTERED writes?  One pool of physical pieces
// (HIP virtual-memory API) created once; a 3.5 GiB virtual range mapped from a chosen subset; a kernel that writes the
// whole range in 1 KiB runs at pseudo-random places (what pb_bin_kernel does to the PageRank value stream), and a plain
// streaming write for comparison.  Subsets: consecutive pieces at every position, strided, random.
// build: hipcc -O3 --offload-arch=gfx950 tools/scatterprobe.hip -o tools/scatterprobe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// every wavefront writes 1 KiB runs (64 lanes x float4); run r of the range goes to place perm(r): an odd multiplier
// modulo the (power of two) number of runs — a bijection, so the whole range is written exactly once
__global__ __launch_bounds__(1024) void scatter_write(float4 *__restrict__ out, uint64_t runs, uint64_t mul, float v)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < runs; r += nwaves) {
        const uint64_t place = (r * mul + 12345u) & (runs - 1);
        out[place * 64 + lane] = make_float4(v, v, v, (float)r);
    }
}
__global__ __launch_bounds__(1024) void stream_write(float4 *__restrict__ out, uint64_t n16, float v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        out[i] = make_float4(v, v, v, v);
}
int main(int argc, char **argv)
{
    const size_t piece_mib = argc > 1 ? atoi(argv[1]) : 256, pool = argc > 2 ? atoi(argv[2]) : 256;
    const size_t piece = piece_mib << 20, need = (size_t)1 << 32 >> 0; // 4 GiB range (power of two number of runs)
    const size_t count = need / piece;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    std::vector<hipMemGenericAllocationHandle_t> h(pool);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (size_t i = 0; i < pool; ++i) CK(hipMemCreate(&h[i], piece, &prop, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%zu pieces of %zu MiB created in %.1f ms; range = %zu pieces\n", pool, piece_mib, ms, count);
    void *va = nullptr; CK(hipMemAddressReserve(&va, need, piece, nullptr, 0));
    hipMemAccessDesc desc{}; desc.location = prop.location; desc.flags = hipMemAccessFlagsProtReadWrite;
    auto measure = [&](const std::vector<size_t> &pick, float *t_scatter, float *t_stream) {
        for (size_t i = 0; i < count; ++i) CK(hipMemMap((char *)va + i * piece, piece, 0, h[pick[i]], 0));
        CK(hipMemSetAccess(va, need, &desc, 1));
        const uint64_t runs = need / 1024;
        float best_s = 1e9f, best_w = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(scatter_write, dim3(512), dim3(1024), 0, 0, (float4 *)va, runs, 0x9E3779B1ull | 1ull, (float)rep);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best_s) best_s = ms;
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(stream_write, dim3(2048), dim3(1024), 0, 0, (float4 *)va, need / 16, (float)rep);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best_w) best_w = ms;
        }
        CK(hipMemUnmap(va, need));
        *t_scatter = best_s; *t_stream = best_w;
    };
    float ts, tw;
    printf("consecutive pieces starting at k (scatter ms / stream ms):\n");
    for (size_t k = 0; k + count <= pool; k += count / 2 ? count / 2 : 1) {
        std::vector<size_t> pick(count);
        for (size_t i = 0; i < count; ++i) pick[i] = k + i;
        measure(pick, &ts, &tw);
        printf("  k=%4zu: %.3f / %.3f\n", k, ts, tw);
    }
    printf("strided subsets (stride s from 0):\n");
    for (size_t s = 2; (count - 1) * s < pool; s += (s < 8 ? 1 : s / 4)) {
        std::vector<size_t> pick(count);
        for (size_t i = 0; i < count; ++i) pick[i] = i * s;
        measure(pick, &ts, &tw);
        printf("  s=%4zu (span %5.1f GiB): %.3f / %.3f\n", s, (double)((count - 1) * s + 1) * piece_mib / 1024, ts, tw);
    }
    printf("half from [0, %zu), half from [j, j + %zu):\n", count / 2, count / 2);
    for (size_t j = count / 2; j + count / 2 <= pool; j += count / 2) {
        std::vector<size_t> pick(count);
        for (size_t i = 0; i < count / 2; ++i) pick[2 * i] = i, pick[2 * i + 1] = j + i;
        measure(pick, &ts, &tw);
        printf("  j=%4zu: %.3f / %.3f\n", j, ts, tw);
    }
    printf("random subsets:\n");
    uint64_t st = 7;
    for (int rep = 0; rep < 12; ++rep) {
        std::vector<size_t> all(pool);
        for (size_t i = 0; i < pool; ++i) all[i] = i;
        for (size_t i = 0; i < count; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; std::swap(all[i], all[i + (st >> 33) % (pool - i)]); }
        all.resize(count);
        measure(all, &ts, &tw);
        printf("  rep %2d: %.3f / %.3f\n", rep, ts, tw);
    }
    return 0;
}
