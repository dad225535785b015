// Experiment: throughput of fql_gemm64_kernel on [M x 512] x [512 x 512] layers, P problems per launch.
#define FQL_STAMPS 1
#include "../fql_amd/csrc/fql_kernels.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 256, N = 512, K = argc > 2 ? atoi(argv[2]) : 512, P = argc > 3 ? atoi(argv[3]) : 7;
    const int flags = argc > 4 ? atoi(argv[4]) : (GF_BIAS | GF_GELU);
    const int RI = argc > 5 ? atoi(argv[5]) : 2;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *A0, *A1, *W, *b, *part; GemmTask* tb;
    CK(hipMalloc(&A0, (size_t)P * M * 512 * 4)); CK(hipMalloc(&A1, (size_t)P * M * 512 * 4)); CK(hipMalloc(&W, (size_t)P * K * N * 4)); CK(hipMalloc(&b, 4096 * 4));
    CK(hipMalloc(&part, (size_t)P * M * 16 * 4));
    CK(hipMemset(A0, 0, (size_t)P * M * 512 * 4)); CK(hipMemset(A1, 0, (size_t)P * M * 512 * 4)); CK(hipMemset(W, 0, (size_t)P * K * N * 4)); CK(hipMemset(b, 0, 4096 * 4));
    CK(hipMemset(part, 0, (size_t)P * M * 16 * 4));
    CK(hipMalloc(&tb, 2 * P * sizeof(GemmTask)));
    std::vector<GemmTask> h(2 * P);
    int grid = 0;
    for (int pp = 0; pp < 2; ++pp) {
        grid = 0;
        for (int i = 0; i < P; ++i) {
            GemmTask t{};
            t.A = (pp ? A1 : A0) + (size_t)i * M * 512; t.C = (pp ? A0 : A1) + (size_t)i * M * 512; t.lda = K; t.ldc = N;
            t.B = W + (size_t)i * K * N; t.ldb = N; t.bias = b; t.M = M; t.N = N; t.K = K;
            t.flags = flags; t.ntn = N / 64; t.tile0 = grid; t.tmt = RI; t.ln_g = b; t.ln_b = b; t.ln_width = K;
            t.aux = part + (size_t)i * M * 16; t.aux2 = part + (size_t)i * M * 16; t.i0 = K / 64; t.i1 = N / 64;
            grid += (M / (32 * RI)) * t.ntn;
            h[pp * P + i] = t;
        }
    }
    CK(hipMemcpy(tb, h.data(), 2 * P * sizeof(GemmTask), hipMemcpyHostToDevice));
    const size_t lds = (size_t)(4 * 64 * 68 + 256) * 4;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 48; ++i) hipLaunchKernelGGL(fql_gemm64_kernel, dim3(grid), dim3(256), lds, s, tb + (i % 2) * P, P);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (30 * 48);
    printf("gemm64 M=%d K=%d P=%d flags=%d grid=%d : %.2f us per launch (%.1f TFLOP/s)\n", M, K, P, flags, grid, us, 2.0 * P * M * N * K / us / 1e6);
    return 0;
}
