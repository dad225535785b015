// Experiment: one forward level of the side lane (lane 1) alone on the chip, as fql_side_kernel runs it: the one-step actor
// on 768 rows plus three 256-row passes, [M x 512] x [512 x 512] each = 384 tiles of 32 x 64.  48 launches per hipGraph.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o experiments/side_bench experiments/side_bench.hip
//         (-DFQL_STAMPS: in-kernel phase stamps; never quote that build's run time)
// argv: [3] = 1: precision = 2 (fql_side_split_kernel, bf16x3 tile body)   [1] flags preset: 0 = bias+gelu+save_z, 1 = + LayerNorm on A and LN partials out (critic layers), 2 = dgrad (W^T) of 4 x 256 rows   [2] tile: 1 = 32x32, 2 = 32x64, 4 = 64x64
#include "../fql_amd/csrc/fql_kernels.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int preset = argc > 1 ? atoi(argv[1]) : 0, shape = argc > 2 ? atoi(argv[2]) : 2;
    const int RI = shape == 4 ? 2 : 1, NJ = shape == 1 ? 1 : 2;
    const int split = argc > 3 ? atoi(argv[3]) : 0;
    const int xg = argc > 4 ? atoi(argv[4]) : 0;   // XCD-aware tile order with xg row groups (0 = row-major)
    const int N = 512, K = 512;
    const int Ms[4] = {preset >= 2 ? 256 : 768, 256, 256, 256};
    int flags = GF_BIAS | GF_GELU | GF_SAVE_Z;
    if (preset == 1) flags |= GF_A_LN | GF_LN_WRITE | GF_LN_PART;
    if (preset == 2) flags = GF_TRANS_B;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t rows = Ms[0] + 3 * 256;
    float *A0, *A1, *Z, *Z2, *XN, *W, *b, *part, *stats;
    CK(hipMalloc(&A0, rows * 512 * 4)); CK(hipMalloc(&A1, rows * 512 * 4)); CK(hipMalloc(&Z, rows * 512 * 4)); CK(hipMalloc(&Z2, rows * 512 * 4)); CK(hipMalloc(&XN, rows * 512 * 4));
    CK(hipMalloc(&W, (size_t)4 * K * N * 4)); CK(hipMalloc(&b, 4096 * 4)); CK(hipMalloc(&part, rows * 32 * 4)); CK(hipMalloc(&stats, rows * 2 * 4));
    std::vector<float> hw((size_t)4 * K * N), ha(rows * 512);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f) * 0.08f;
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = ((float)((i * 40503u) & 0xFFFF) / 65536.0f - 0.5f);
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(A0, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(A1, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ones(4096, 1.0f);
    CK(hipMemcpy(b, ones.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(part, 0, rows * 32 * 4)); CK(hipMemset(stats, 0, rows * 2 * 4)); CK(hipMemcpy(Z, ha.data(), rows * 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(XN, ha.data(), rows * 512 * 4, hipMemcpyHostToDevice));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 8 * 4096 * 8)); CK(hipMemset(stamps, 0, 8 * 4096 * 8));
    GemmTask* tb; CK(hipMalloc(&tb, 8 * sizeof(GemmTask)));
    std::vector<GemmTask> h(8);
    int grid = 0;
    for (int pp = 0; pp < 2; ++pp) {
        grid = 0;
        size_t r0 = 0;
        for (int i = 0; i < 4; ++i) {
            GemmTask t{};
            t.A = (pp ? A1 : A0) + r0 * 512; t.C = (pp ? A0 : A1) + r0 * 512; t.Zout = Z + r0 * 512; t.lda = K; t.ldc = N;
            t.B = W + (size_t)i * K * N; t.ldb = N; t.bias = b + 512; t.M = Ms[i]; t.N = N; t.K = K;
            t.flags = flags; t.ntn = N / (32 * NJ); t.tile0 = grid; t.tmt = RI; t.wk = NJ; t.xg = xg;
            t.ln_g = b; t.ln_b = b + 1024; t.ln_width = K; t.ln_xout = XN + r0 * 512; t.ln_stats = stats + r0 * 2;
            t.aux = part + r0 * 32; t.aux2 = part + r0 * 32; t.i0 = K / 32; t.i1 = N / 32;
#ifdef FQL_STAMPS
            t.stamps64 = stamps;
#endif
            grid += (Ms[i] / (32 * RI)) * t.ntn;
            r0 += Ms[i];
            h[pp * 4 + i] = t;
        }
    }
    CK(hipMemcpy(tb, h.data(), 8 * sizeof(GemmTask), hipMemcpyHostToDevice));
    const size_t lds = split ? (size_t)FQL_TILE_SPLIT_LDS_FLOATS(NJ) * 4 : (size_t)(shape == 4 ? 2 * (64 + 64) * 68 + 256 : shape == 1 ? 2 * 32 * 68 + 2 * 64 * 36 + 128 : 2 * 32 * 68 + 2 * 64 * 68 + 128) * 4;
    
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 48; ++i)
        if (shape == 4) hipLaunchKernelGGL(fql_side_big_kernel, dim3(grid), dim3(256), lds, s, (const GemmTask*)(tb + (i % 2) * 4), 4, (const WgradTask*)nullptr, 0,
                           (const LnBwdTask*)nullptr, 0, grid, grid, (const MiscTask*)nullptr, grid, 0, -1);
        else if (split) hipLaunchKernelGGL(fql_side_split_kernel, dim3(grid), dim3(256), lds, s, (const GemmTask*)(tb + (i % 2) * 4), 4, (const WgradTask*)nullptr, 0,
                           (const LnBwdTask*)nullptr, 0, grid, grid, (const MiscTask*)nullptr, grid, 0, -1);
        else hipLaunchKernelGGL(fql_side_kernel, dim3(grid), dim3(256), lds, s, (const GemmTask*)(tb + (i % 2) * 4), 4, (const WgradTask*)nullptr, 0,
                           (const LnBwdTask*)nullptr, 0, grid, grid, (const MiscTask*)nullptr, grid, 0, -1);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (30 * 48);
    const double flop = 2.0 * rows * N * K;
    printf("side level %s xg=%d preset=%d tile=%d grid=%d : %.2f us per launch (%.1f TFLOP/s; MFMA-bound floor %.2f us)\n", split ? "bf16x3" : "fp32", xg, preset, shape, grid, us, flop / us / 1e6,
           flop / 157.3e6);
#ifdef FQL_STAMPS
    {
        std::vector<unsigned long long> st(8 * grid);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        double d[8] = {0};
        std::vector<double> entry(grid), exitt(grid);
        for (int w = 0; w < grid; ++w) tmin = std::min(tmin, st[8 * w]);
        for (int w = 0; w < grid; ++w) {
            for (int i = 0; i < 5; ++i) { tmax = std::max(tmax, st[8 * w + i]); if (i) d[i] += (double)(st[8 * w + i] - st[8 * w + i - 1]); }
            entry[w] = (double)(st[8 * w] - tmin) / 100; exitt[w] = (double)(st[8 * w + 4] - tmin) / 100;
        }
        std::sort(entry.begin(), entry.end()); std::sort(exitt.begin(), exitt.end());
        printf("   in-kernel us (mean over WGs): task lookup + LN stats %.2f | first chunk staged %.2f | K loop %.2f | epilogue %.2f\n",
               d[1] / grid / 100, d[2] / grid / 100, d[3] / grid / 100, d[4] / grid / 100);
        printf("   WG entry (us after the first): median %.2f  p90 %.2f  max %.2f ; WG exit: median %.2f  p90 %.2f  max %.2f\n", entry[grid / 2],
               entry[grid * 9 / 10], entry[grid - 1], exitt[grid / 2], exitt[grid * 9 / 10], exitt[grid - 1]);
    }
#endif
    return 0;
}
