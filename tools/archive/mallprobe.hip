// mallprobe.hip — does the 256 MB memory-side cache (MALL / Infinity Cache) of an MI355X keep a freshly written
// stream for a consumer kernel?  Producer kernel writes S bytes, consumer kernel reads them; both timed over a
// range of S.  (PageRank's value stream is written by pb_bin_kernel and read by pb_accum_kernel: 3.6 GB per sweep
// at scale 26 — could a sweep cut into row super-blocks keep that traffic out of HBM?)
// build: hipcc -O3 --offload-arch=gfx950 tools/mallprobe.hip -o tools/mallprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void producer(float4 *__restrict__ out, uint64_t n16, float v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        out[i] = make_float4(v, v + 1.f, v + 2.f, (float)i);
}
__global__ __launch_bounds__(256) void consumer(const float4 *__restrict__ in, uint64_t n16, float *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *out = acc;
}
int main()
{
    float4 *buf; float *out;
    const uint64_t maxb = 4ull << 30;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, maxb));
    hipEvent_t a, b, c;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    const uint64_t sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096};
    for (uint64_t mb : sizes_mb) {
        const uint64_t bytes = mb << 20, n16 = bytes / 16;
        const int iters = mb <= 256 ? 40 : 8;
        double tw = 0, tr = 0;
        for (int it = -2; it < iters; ++it) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(producer, dim3(256 * 8), dim3(256), 0, 0, buf, n16, (float)it);
            CK(hipEventRecord(b));
            hipLaunchKernelGGL(consumer, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)buf, n16, out);
            CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
            float w, r; CK(hipEventElapsedTime(&w, a, b)); CK(hipEventElapsedTime(&r, b, c));
            if (it >= 0) { tw += w; tr += r; }
        }
        printf("S = %5llu MiB: write %8.1f GB/s   read-after-write %8.1f GB/s\n", (unsigned long long)mb, bytes * iters / tw / 1e6,
               bytes * iters / tr / 1e6);
    }
    return 0;
}
