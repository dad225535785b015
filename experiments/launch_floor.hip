// Micro-benchmark: per-launch cost of dependent kernels on one stream (eager and hipGraph).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
__global__ void k_chain3(float* p) {  // 3 dependent global round trips
    if (threadIdx.x == 0 && blockIdx.x == 0) { int i = (int)p[0]; int j = (int)p[64 + i]; p[128 + j] += 1.0f; }
}
__global__ void k_wide(float* p, int n) {  // 256 WGs, each reads+writes its own line
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
template <typename F> double time_launches(hipStream_t s, int n, F f) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < n; ++i) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* p; CK(hipMalloc(&p, 1 << 22)); CK(hipMemset(p, 0, 1 << 22));
    const int N = 2000;
    printf("eager  empty<<<1,64>>>      %.2f us\n", time_launches(s, N, [&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, s); }));
    printf("eager  empty<<<256,256>>>   %.2f us\n", time_launches(s, N, [&] { hipLaunchKernelGGL(k_empty, 256, 256, 0, s); }));
    printf("eager  touch<<<1,64>>>      %.2f us\n", time_launches(s, N, [&] { hipLaunchKernelGGL(k_touch, 1, 64, 0, s, p); }));
    printf("eager  chain3<<<1,64>>>     %.2f us\n", time_launches(s, N, [&] { hipLaunchKernelGGL(k_chain3, 1, 64, 0, s, p); }));
    printf("eager  wide<<<256,256>>>    %.2f us\n", time_launches(s, N, [&] { hipLaunchKernelGGL(k_wide, 256, 256, 0, s, p, 65536); }));
    for (int variant = 0; variant < 4; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) {
            if (variant == 0) hipLaunchKernelGGL(k_empty, 1, 64, 0, s);
            if (variant == 1) hipLaunchKernelGGL(k_empty, 256, 256, 0, s);
            if (variant == 2) hipLaunchKernelGGL(k_chain3, 1, 64, 0, s, p);
            if (variant == 3) hipLaunchKernelGGL(k_wide, 256, 256, 0, s, p, 65536);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double t = time_launches(s, 100, [&] { hipGraphLaunch(ge, s); });
        const char* nm[] = {"empty<<<1,64>>>", "empty<<<256,256>>>", "chain3<<<1,64>>>", "wide<<<256,256>>>"};
        printf("graph  %-20s %.2f us per kernel (100 per graph)\n", nm[variant], t / 100);
    }
    return 0;
}
