// Experiment: per-launch time of the Euler chain's launches, alone on the chip, 96 launches per hipGraph:
//   old  = fql_gemm16_kernel on one 256x512x512 layer (the round-1 chain launch)
//   new  = fql_chain_kernel variants B (middle layer), and the A/B/C triple of one Euler step
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o experiments/chain_bench experiments/chain_bench.hip
//         (add -DFQL_STAMPS for the in-kernel phase stamps; never quote that build's run time)
#include "../fql_amd/csrc/fql_chain.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static void launch_chain(hipStream_t s, const ChainArgs& a) {
    if (a.variant == 0) hipLaunchKernelGGL((fql_chain_kernel<512, 0>), dim3(256), dim3(512), FQL_CHAIN_LDS_BYTES(512), s, a);
    else if (a.variant == 1) hipLaunchKernelGGL((fql_chain_kernel<512, 1>), dim3(256), dim3(512), FQL_CHAIN_LDS_BYTES(512), s, a);
    else hipLaunchKernelGGL((fql_chain_kernel<512, 2>), dim3(256), dim3(512), FQL_CHAIN_LDS_BYTES(512), s, a);
}
static double time_graph(hipStream_t s, hipGraphExec_t ge, int launches) {
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (50.0 * launches);
}

int main(int argc, char** argv) {
    const int M = 256, H = 512, ap = 16, ad = 8;
    const int prio = argc > 1 ? atoi(argv[1]) : 0;
    (void)prio;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *A0, *A1, *W, *Wf, *b, *C0f, *W0f, *W4f, *evp, *ea0, *ea1;
    CK(hipMalloc(&A0, (size_t)M * H * 4)); CK(hipMalloc(&A1, (size_t)M * H * 4));
    CK(hipMalloc(&W, (size_t)8 * H * H * 4)); CK(hipMalloc(&Wf, (size_t)8 * H * H * 4)); CK(hipMalloc(&b, 4096 * 4));
    CK(hipMalloc(&C0f, (size_t)M * H * 4)); CK(hipMalloc(&W0f, (size_t)16 * H * 4)); CK(hipMalloc(&W4f, (size_t)H * ap * 4));
    CK(hipMalloc(&evp, (size_t)(H / 32) * M * ap * 4)); CK(hipMalloc(&ea0, (size_t)M * 64 * 4)); CK(hipMalloc(&ea1, (size_t)M * ap * 4));
    std::vector<float> hw((size_t)8 * H * H);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f) * 0.08f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wf, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ha((size_t)M * H);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = ((float)((i * 40503u) & 0xFFFF) / 65536.0f - 0.5f);
    CK(hipMemcpy(A0, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(A1, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(C0f, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W0f, hw.data(), (size_t)16 * H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W4f, hw.data(), (size_t)H * ap * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b, 0, 4096 * 4)); CK(hipMemset(evp, 0, (size_t)(H / 32) * M * ap * 4)); CK(hipMemset(ea0, 0, (size_t)M * 64 * 4)); CK(hipMemset(ea1, 0, (size_t)M * ap * 4));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 8 * 4096 * 8)); CK(hipMemset(stamps, 0, 8 * 4096 * 8));
    CK(hipFuncSetAttribute((const void*)fql_chain_kernel<512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FQL_CHAIN_LDS_BYTES(512)));

    // ---- old: gemm16 kernel, 8 tasks ping-pong with different weights
    {
        GemmTask* tb; CK(hipMalloc(&tb, 16 * sizeof(GemmTask)));
        std::vector<GemmTask> h(8);
        int grid = 0;
        for (int i = 0; i < 8; ++i) {
            GemmTask t{};
            t.A = (i & 1) ? A1 : A0; t.C = (i & 1) ? A0 : A1; t.lda = H; t.ldc = H;
            t.B = W + (size_t)i * H * H; t.ldb = H; t.bias = b; t.M = M; t.N = H; t.K = H;
            t.flags = GF_BIAS | GF_GELU; t.wk = 2; t.ntn = (H / 16 + 1) / 2; t.tile0 = 0; t.tmt = 1;
#ifdef FQL_STAMPS
            t.aux = (float*)stamps;
#endif
            grid = (M / 16) * t.ntn;
            h[i] = t;
        }
        CK(hipMemcpy(tb, h.data(), 8 * sizeof(GemmTask), hipMemcpyHostToDevice));
        const size_t lds = ((size_t)16 * (H + 4) + 1024 + 1280) * 4;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 96; ++i) hipLaunchKernelGGL((fql_gemm16_kernel<false, false>), dim3(grid), dim3(256), lds, s, tb + (i % 8), 1, -1);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("old gemm16 layer (256 threads, row-major W, table task): %.2f us per launch\n", time_graph(s, ge, 96));
    }
    auto base = [&](int i) {
        ChainArgs a{};
        a.A = (i & 1) ? A1 : A0; a.C = (i & 1) ? A0 : A1; a.Wf = Wf + (size_t)(i % 8) * H * H; a.bias = b;
        a.M = M; a.ad = ad; a.ap = ap; a.inv_steps = 0.1f; a.t_s = 0.3f; a.variant = 1; a.tl = -1;
#ifdef FQL_STAMPS
        a.stamps = stamps;
#endif
        return a;
    };
    auto report_stamps = [&](const char* what) {
#ifdef FQL_STAMPS
        const int grid = 256;
        std::vector<unsigned long long> st(8 * grid);
        CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        double d[8] = {0};
        for (int w = 0; w < grid; ++w) {
            tmin = std::min(tmin, st[8 * w]);
            for (int i = 0; i < 5; ++i) { tmax = std::max(tmax, st[8 * w + i]); if (i) d[i] += (double)(st[8 * w + i] - st[8 * w + i - 1]); }
        }
        double spread = 0; for (int w = 0; w < grid; ++w) spread = std::max(spread, (double)(st[8 * w] - tmin) / 100);
        printf("   %s in-kernel us (mean over WGs): loads/stage %.2f | barrier %.2f | MFMA %.2f | reduce+epilogue %.2f ; first entry..last exit %.2f ; entry spread %.2f\n",
               what, d[1] / grid / 100, d[2] / grid / 100, d[3] / grid / 100, d[4] / grid / 100, (double)(tmax - tmin) / 100, spread);
#else
        (void)what;
#endif
    };
    // ---- new: variant B only
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 96; ++i) { ChainArgs a = base(i); launch_chain(s, a); }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("new chain kernel, variant B (512 threads, fragment-major W, kernarg task): %.2f us per launch\n", time_graph(s, ge, 96));
        report_stamps("B");
    }
    // ---- new: one Euler step = A, B, C
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 96; ++i) {
            ChainArgs a = base(i);
            const int v = i % 3;
            a.variant = v;
            if (v == 0) { a.A = C0f; a.W0f = W0f; a.ea_in = ea0; a.ea_ld = 64; a.ea_out = ea1; a.evp_in = evp; a.eb = b; a.C = A0; }
            if (v == 1) { a.A = A0; a.C = A1; }
            if (v == 2) { a.A = A1; a.C = nullptr; a.W4f = W4f; a.evp_out = evp; }
            launch_chain(s, a);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double us = time_graph(s, ge, 96);
        printf("new chain kernel, Euler step A+B+C: %.2f us per launch = %.2f us per step (round 1: 6.9 + 10.7 + 7.0 = 24.6)\n", us, 3 * us);
    }
    for (int v = 0; v < 3; v += 2) {   // variants A and C alone
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 96; ++i) {
            ChainArgs a = base(i);
            a.variant = v;
            if (v == 0) { a.A = C0f; a.W0f = W0f; a.ea_in = ea0; a.ea_ld = 64; a.ea_out = ea1; a.evp_in = evp; a.eb = b; }
            if (v == 2) { a.C = nullptr; a.W4f = W4f; a.evp_out = evp; }
            launch_chain(s, a);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("new chain kernel, variant %c alone: %.2f us per launch\n", v == 0 ? 'A' : 'C', time_graph(s, ge, 96));
        report_stamps(v == 0 ? "A" : "C");
    }
    return 0;
}
