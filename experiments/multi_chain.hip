// Experiment: how well do N independent latency-bound kernel chains overlap on N streams (graph branches)?
#include "../fql_amd/csrc/fql_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
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
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 256;
    const int NS = 4, NL = 48;
    hipStream_t s[NS]; hipEvent_t ef, ej[NS];
    for (int i = 0; i < NS; ++i) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming)); }
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    Chain c[NS];
    for (int i = 0; i < NS; ++i) c[i] = make(M, 512, 512);
    for (int ns = 1; ns <= NS; ++ns) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, s[0]));
        for (int i = 1; i < ns; ++i) CK(hipStreamWaitEvent(s[i], ef, 0));
        for (int l = 0; l < NL; ++l)
            for (int i = 0; i < ns; ++i)
                hipLaunchKernelGGL((fql_gemm16_kernel<false, false>), dim3(c[i].grid), dim3(256), c[i].lds, s[i], c[i].tb + (l % 8), 1);
        for (int i = 1; i < ns; ++i) { CK(hipEventRecord(ej[i], s[i])); CK(hipStreamWaitEvent(s[0], ej[i], 0)); }
        CK(hipStreamEndCapture(s[0], &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 30;
        printf("M=%d: %d concurrent chains of %d layers: %.1f us total, %.2f us per layer per chain, aggregate %.2f us per layer\n", M, ns, NL, us, us / NL, us / NL / ns);
    }
    return 0;
}
