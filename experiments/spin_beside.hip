// Experiment: does a resident, idle (sleeping) kernel on every CU slow an independent kernel chain on another stream?
#include "../fql_amd/csrc/fql_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int NREG>
__global__ __launch_bounds__(256) void k_spin(unsigned long long ticks, float* sink, int poll, unsigned* flag) {
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (poll) { if (__hip_atomic_load((const FQL_GAS unsigned*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 12345u) break; }
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
}
struct Chain { float *A0, *A1, *W, *b; GemmTask* tb; int grid; size_t lds; };
static Chain make(int M, int N, int K) {
    Chain c;
    CK(hipMalloc(&c.A0, (size_t)M * 1024 * 4)); CK(hipMalloc(&c.A1, (size_t)M * 1024 * 4)); CK(hipMalloc(&c.W, (size_t)8 * K * N * 4)); CK(hipMalloc(&c.b, 4096 * 4));
    CK(hipMemset(c.A0, 0, (size_t)M * 1024 * 4)); CK(hipMemset(c.A1, 0, (size_t)M * 1024 * 4)); CK(hipMemset(c.W, 0, (size_t)8 * K * N * 4)); CK(hipMemset(c.b, 0, 4096 * 4));
    CK(hipMalloc(&c.tb, 16 * sizeof(GemmTask)));
    std::vector<GemmTask> h(8);
    for (int i = 0; i < 8; ++i) {
        GemmTask t{};
        t.A = (i & 1) ? c.A1 : c.A0; t.C = (i & 1) ? c.A0 : c.A1; t.lda = K; t.ldc = N;
        t.B = c.W + (size_t)i * K * N; t.ldb = N; t.bias = c.b; t.M = M; t.N = N; t.K = K;
        t.flags = GF_BIAS | GF_GELU; t.wk = 2; t.tmt = 1; t.ntn = N / 32; t.tile0 = 0;
        c.grid = (M / 16) * t.ntn;
        h[i] = t;
    }
    CK(hipMemcpy(c.tb, h.data(), 8 * sizeof(GemmTask), hipMemcpyHostToDevice));
    c.lds = ((size_t)16 * (K + 4) + 1024 + 1280) * 4;
    return c;
}
int main() {
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    Chain c = make(256, 512, 512);
    float* sink; CK(hipMalloc(&sink, 64)); unsigned* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
    const int NL = 48;
    for (int variant = 0; variant < 5; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
        const unsigned long long ticks = 25000;  // 250 us
        if (variant == 1) hipLaunchKernelGGL((k_spin<8>), dim3(256), dim3(256), 0, s1, ticks, sink, 0, flag);
        if (variant == 2) hipLaunchKernelGGL((k_spin<200>), dim3(256), dim3(256), 0, s1, ticks, sink, 0, flag);
        if (variant == 3) hipLaunchKernelGGL((k_spin<8>), dim3(256), dim3(256), 40000, s1, ticks, sink, 0, flag);
        if (variant == 4) hipLaunchKernelGGL((k_spin<8>), dim3(256), dim3(256), 0, s1, ticks, sink, 1, flag);
        for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((fql_gemm16_kernel<false, false>), dim3(c.grid), dim3(256), c.lds, s0, c.tb + (l % 8), 1);
        CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0));
        CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s0));
        CK(hipStreamSynchronize(s0));
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 20;
        const char* nm[] = {"chain alone", "beside 250us sleeper (8 regs)", "beside 250us sleeper (200 regs)", "beside 250us sleeper (8 regs, 40 KB LDS)", "beside 250us sleeper polling a flag"};
        printf("%-45s: %.1f us\n", nm[variant], us);
    }
    return 0;
}
