// One float convolution launch alone on the chip (fql_conv3x3_kernel / fql_conv3x3_split_kernel), as the visual update issues it:
// N images of H x W x Ci -> Co, one workgroup per row block.  -DFQL_STAMPS: in-kernel phases of the first row block of every workgroup.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DFQL_STAMPS] -o experiments/conv_bench experiments/conv_bench.hip
// argv: [1] H (= W)  [2] Ci  [3] Co  [4] images  [5] 1 = split (bf16x3) kernel
#include "../fql_amd/csrc/fql_conv.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 32, W = H, Ci = argc > 2 ? atoi(argv[2]) : 16, Co = argc > 3 ? atoi(argv[3]) : 16;
    const int N = argc > 4 ? atoi(argv[4]) : 1280, split = argc > 5 ? atoi(argv[5]) : 0;
    int R = std::min(H, std::max(1, 128 / W));
    for (; R >= 1; R >>= 1) if (((size_t)(R + 2) * (W + 2) * (Ci + 4) + (size_t)Co * (9 * Ci + 4)) * 4 <= 65536) break;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t nin = (size_t)N * H * W * Ci, nout = (size_t)N * H * W * Co;
    float *in, *out, *outr, *Wl, *Ws, *bias, *K;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&outr, nout * 4));
    CK(hipMalloc(&K, (size_t)9 * Ci * Co * 4)); CK(hipMalloc(&Wl, (size_t)Co * (9 * Ci + 4) * 4 * 2)); CK(hipMalloc(&Ws, (size_t)Co * (9 * Ci + 4) * 4 * 2)); CK(hipMalloc(&bias, 256));
    CK(hipMemset(Wl, 0, (size_t)Co * (9 * Ci + 4) * 8)); CK(hipMemset(Ws, 0, (size_t)Co * (9 * Ci + 4) * 8)); CK(hipMemset(bias, 0, 256));
    std::vector<float> h(nin);
    unsigned st = 12345u;
    for (size_t i = 0; i < nin; ++i) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) / 16777216.0f - 0.5f; }
    CK(hipMemcpy(in, h.data(), nin * 4, hipMemcpyHostToDevice));
    std::vector<float> hk((size_t)9 * Ci * Co);
    for (size_t i = 0; i < hk.size(); ++i) { st = st * 1664525u + 1013904223u; hk[i] = ((float)(st >> 8) / 16777216.0f - 0.5f) * 0.1f; }
    CK(hipMemcpy(K, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
    ConvWprepTask wt[2] = {{K, Wl, nullptr, Ci, Co, Ci, 0}, {K, Ws, nullptr, Ci, Co, Ci, 1}};
    ConvWprepTask* dwt; CK(hipMalloc(&dwt, sizeof(wt))); CK(hipMemcpy(dwt, wt, sizeof(wt), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fql_conv_wprep_kernel, dim3(4, 2), dim3(FQL_THREADS), 0, s, (const ConvWprepTask*)dwt);
    CK(hipStreamSynchronize(s));
    unsigned long long* stamps = nullptr;
    const int grid = N * (H / R);
#ifdef FQL_STAMPS
    CK(hipMalloc(&stamps, (size_t)grid * 64)); CK(hipMemset(stamps, 0, (size_t)grid * 64));
#endif
    ConvArgs a{};
    a.in = in; a.Wl = split ? Ws : Wl; a.bias = bias; a.out = out; a.out_relu = outr; a.mask = nullptr; a.add = nullptr;
    a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Ci_real = Ci; a.Co = Co; a.in_mode = 1; a.transposed = 0; a.R = R; a.tile0 = 0; a.nwg = grid;
#ifdef FQL_STAMPS
    a.stamps = stamps;
#endif
    ConvArgs* da; CK(hipMalloc(&da, sizeof(a))); CK(hipMemcpy(da, &a, sizeof(a), hipMemcpyHostToDevice));
    const size_t lds = split ? (size_t)FQL_CONV_SPLIT_LDS_WORDS(R, W, Ci, Co) * 4 : ((size_t)(R + 2) * (W + 2) * (Ci + 4) + (size_t)Co * (9 * Ci + 4)) * 4;
    auto launch = [&]() {
        if (split) hipLaunchKernelGGL(fql_conv3x3_split_kernel, dim3(grid), dim3(FQL_THREADS), lds, s, (const ConvArgs*)da, 1);
        else hipLaunchKernelGGL(fql_conv3x3_kernel, dim3(grid), dim3(FQL_THREADS), lds, s, (const ConvArgs*)da, 1);
    };
    for (int i = 0; i < 5; ++i) launch();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
    const double flop = 2.0 * N * H * W * 9.0 * Ci * Co;
    printf("conv %s %dx%d %d->%d, %d images, R=%d, grid=%d, lds=%zu B: %.1f us per launch (%.1f TFLOP/s; fp32 MFMA floor %.1f us)\n", split ? "bf16x3" : "fp32", H, W, Ci, Co, N, R,
           grid, lds, us, flop / us / 1e6, flop / 157.3e6);
#ifdef FQL_STAMPS
    {
        std::vector<unsigned long long> v((size_t)grid * 8);
        CK(hipMemcpy(v.data(), stamps, v.size() * 8, hipMemcpyDeviceToHost));
        double d[8] = {0};
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < grid; ++w) { tmin = std::min(tmin, v[8 * (size_t)w]); tmax = std::max(tmax, v[8 * (size_t)w + 4]); for (int i = 1; i < 5; ++i) d[i] += (double)(v[8 * (size_t)w + i] - v[8 * (size_t)w + i - 1]); }
        printf("   per workgroup (mean, us): weights -> LDS %.2f | input rows staged %.2f | MFMA loop %.2f | epilogue %.2f ; launch span %.1f us\n", d[1] / grid / 100, d[2] / grid / 100,
               d[3] / grid / 100, d[4] / grid / 100, (double)(tmax - tmin) / 100);
    }
#endif
    return 0;
}
