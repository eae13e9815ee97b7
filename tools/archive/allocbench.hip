// allocbench — what hipMalloc / hipFree / hipMemset cost on this box (plan construction allocates ~30 GB in ~25 pieces)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    for (size_t mb : {1, 16, 256, 1024, 4096, 8192}) {
        for (int rep = 0; rep < 2; ++rep) {
            void *p = nullptr;
            double t0 = now();
            hipMalloc(&p, mb << 20);
            double t1 = now();
            hipMemset(p, 0, mb << 20);
            hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("%6zu MiB: malloc %.3f ms, first memset %.3f ms, free %.3f ms\n", mb, t1 - t0, t2 - t1, t3 - t2);
        }
    }
    // many medium allocations in a row, as the plan builder does
    std::vector<void *> ps(16);
    double t0 = now();
    for (auto &p : ps)
        hipMalloc(&p, 256u << 20);
    double t1 = now();
    for (auto &p : ps)
        hipFree(p);
    double t2 = now();
    printf("16 x 256 MiB: malloc %.3f ms, free %.3f ms\n", t1 - t0, t2 - t1);
    return 0;
}
