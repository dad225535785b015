// Experiment: do two hipGraph branches (captured via event fork/join) overlap on MI355X?
// lane A: 48 dependent chain layers (M=256); lane B: 24 dependent wide layers (M=1536).
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
    c.lds = ((size_t)16 * (K + 4) + 1024) * 4;
    return c;
}
static double run(hipStream_t s, hipGraphExec_t ge) {
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 50;
}
int main() {
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    Chain a = make(256, 512, 512), b = make(1536, 512, 512);
    const int NA = 48, NB = 24;
    auto laneA = [&](hipStream_t st) { for (int i = 0; i < NA; ++i) hipLaunchKernelGGL((fql_gemm16_kernel<true, false>), dim3(a.grid), dim3(256), a.lds, st, a.tb + (i % 8), 1); };
    auto laneB = [&](hipStream_t st) { for (int i = 0; i < NB; ++i) hipLaunchKernelGGL((fql_gemm16_kernel<true, false>), dim3(b.grid), dim3(256), b.lds, st, b.tb + (i % 8), 1); };
    hipGraph_t g; hipGraphExec_t gA, gB, gSeq, gPar;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); laneA(s); CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gA, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); laneB(s); CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gB, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); laneA(s); laneB(s); CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gSeq, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
    laneA(s); laneB(s2);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0));
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gPar, g, nullptr, nullptr, 0));
    printf("lane A alone (48 chain layers M=256): %.1f us\n", run(s, gA));
    printf("lane B alone (24 wide layers M=1536): %.1f us\n", run(s, gB));
    printf("A then B sequential in one graph:     %.1f us\n", run(s, gSeq));
    printf("A || B forked in one graph:           %.1f us\n", run(s, gPar));
    // eager two streams
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < 20; ++i) { laneA(s); laneB(s2); }
        CK(hipDeviceSynchronize());
        printf("eager two streams:                    %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 20);
    }
    return 0;
}
