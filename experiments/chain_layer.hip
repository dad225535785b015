// Experiment: per-launch time of one Euler-chain layer ([256x512]x[512x512] + bias + GELU) captured
// 100x in a hipGraph, using the engine's own kernels.  Variants via argv.
#ifndef NOSTAMP
#define FQL_STAMPS 1
#endif
#include "../fql_amd/csrc/fql_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 256, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
    const int wk = argc > 4 ? atoi(argv[4]) : 2;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    const int tmt = argc > 6 ? atoi(argv[6]) : 1;  // 1: A never written (C to a third buffer), 2: same W for all layers
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *A0, *A1, *W, *b; GemmTask* tb;
    CK(hipMalloc(&A0, (size_t)M * 1024 * 4)); CK(hipMalloc(&A1, (size_t)M * 1024 * 4)); CK(hipMalloc(&W, (size_t)8 * K * N * 4)); CK(hipMalloc(&b, 4096 * 4));
    CK(hipMemset(A0, 0, (size_t)M * 1024 * 4)); CK(hipMemset(A1, 0, (size_t)M * 1024 * 4)); CK(hipMemset(W, 0, (size_t)8 * K * N * 4)); CK(hipMemset(b, 0, 4096 * 4));
    CK(hipMalloc(&tb, 16 * sizeof(GemmTask)));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 8 * 4096 * 8)); CK(hipMemset(stamps, 0, 8 * 4096 * 8));
    std::vector<GemmTask> h(8);
    int grid = 0;
    for (int i = 0; i < 8; ++i) {   // 8 tasks: ping-pong, different weight matrices (as the 4 hidden layers of 2 steps)
        GemmTask t{};
        t.A = (i & 1) ? A1 : A0; t.C = (i & 1) ? A0 : A1; t.lda = K; t.ldc = N;
        if (mode & 1) { t.A = A0; t.C = A1; }
        t.B = W + (size_t)((mode & 2) ? 0 : i) * K * N; t.ldb = N; t.bias = b; t.M = M; t.N = N; t.K = K;
        t.flags = GF_BIAS | GF_GELU; t.aux = (float*)stamps; t.wk = wk; t.ntn = (N / 16 + (4 / wk) - 1) / (4 / wk); t.tile0 = 0;
        t.tmt = tmt; grid = (M / (16 * tmt)) * t.ntn;
        if (mode & 4) { t.flags |= GF_TRANS_B; t.ldb = K; }  // B read as W^T [N][K]: 16-byte fragment loads
        h[i] = t;
    }
    CK(hipMemcpy(tb, h.data(), 8 * sizeof(GemmTask), hipMemcpyHostToDevice));
    const size_t lds = ((size_t)16 * tmt * (K + 4) + 1024 * tmt) * 4;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 96; ++i) hipLaunchKernelGGL((fql_gemm16_kernel<true, false>), dim3(grid), dim3(256), lds, s, tb + (i % 8), 1, -1);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (50 * 96);
    printf("mode=%d M=%d N=%d K=%d wk=%d grid=%d : %.2f us per layer launch (%.1f TFLOP/s)\n", mode, M, N, K, wk, grid, us, 2.0 * M * N * K / us / 1e6);
#ifdef FQL_STAMPS
    {   // stamps of the last launch: per-WG deltas (10 ns ticks), averaged
        std::vector<unsigned long long> st(8 * grid);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        double d[8] = {0};
        for (int w = 0; w < grid; ++w) {
            tmin = std::min(tmin, st[8 * w]);
            for (int i = 0; i < 6; ++i) { if (st[8 * w + i] > tmax) tmax = st[8 * w + i]; if (i) d[i] += (double)(st[8 * w + i] - st[8 * w + i - 1]); }
        }
        printf("   in-kernel (us, mean over WGs): A-load %.2f | barrier %.2f | B+MFMA %.2f | reduce %.2f | epilogue %.2f ; first-entry..last-exit %.2f\n",
               d[1] / grid / 100, d[2] / grid / 100, d[3] / grid / 100, d[4] / grid / 100, d[5] / grid / 100, (double)(tmax - tmin) / 100);
        double spread = 0; for (int w = 0; w < grid; ++w) spread = std::max(spread, (double)(st[8 * w] - tmin) / 100);
        printf("   WG entry spread %.2f us\n", spread);
    }
#endif
    return 0;
}
