// Check + timing of fql_chain_split_kernel (precision = 2) against fql_chain_kernel (fp32) on the same inputs, variants A / B / C.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o experiments/chain_split_check experiments/chain_split_check.hip
#include "../fql_amd/csrc/fql_chain.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dev(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((float)(s >> 8) / 16777216.0f - 0.5f) * scale; }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
static std::vector<float> host(const float* d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }
static double maxdiff(const std::vector<float>& a, const std::vector<float>& b, double* ref) {
    double m = 0, r = 0;
    for (size_t i = 0; i < a.size(); ++i) { m = std::max(m, (double)std::fabs(a[i] - b[i])); r = std::max(r, (double)std::fabs(a[i])); }
    *ref = r; return m;
}

// host bf16 (round to nearest even) split of an [M][H] tensor into the hi / lo word planes the split chain passes between launches
static unsigned bf16_rne(float x) { unsigned u; memcpy(&u, &x, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; }
static float bf16_f(unsigned b) { unsigned u = b << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<unsigned> split_planes(const std::vector<float>& a, int M, int H) {
    std::vector<unsigned> w((size_t)M * H);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < H; k += 2) {
            unsigned h[2], l[2];
            for (int e = 0; e < 2; ++e) { const float x = a[(size_t)m * H + k + e]; h[e] = bf16_rne(x); l[e] = bf16_rne(x - bf16_f(h[e])); }
            w[(size_t)m * (H / 2) + k / 2] = h[0] | (h[1] << 16);
            w[(size_t)M * (H / 2) + (size_t)m * (H / 2) + k / 2] = l[0] | (l[1] << 16);
        }
    return w;
}
static std::vector<float> join_planes(const std::vector<float>& wf, int M, int H) {
    std::vector<float> a((size_t)M * H);
    const unsigned* w = reinterpret_cast<const unsigned*>(wf.data());
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < H; ++k) {
            const unsigned hw = w[(size_t)m * (H / 2) + k / 2], lw = w[(size_t)M * (H / 2) + (size_t)m * (H / 2) + k / 2];
            a[(size_t)m * H + k] = bf16_f((k & 1) ? hw >> 16 : hw & 0xFFFFu) + bf16_f((k & 1) ? lw >> 16 : lw & 0xFFFFu);
        }
    return a;
}

int main() {
    const int M = 256, H = 512, ap = 16, ad = 8, od = 29;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* A = dev((size_t)M * H, 1, 1.0f);          // activations in
    float* As;                                        // the same, as hi / lo planes
    { std::vector<unsigned> w = split_planes(host(A, (size_t)M * H), M, H); CK(hipMalloc(&As, (size_t)M * H * 4)); CK(hipMemcpy(As, w.data(), (size_t)M * H * 4, hipMemcpyHostToDevice)); }
    float* C0r = dev((size_t)M * H, 2, 1.0f);        // loop-invariant layer-0 part, row-major (what the split kernel reads)
    float* C0;                                        // the same in accumulator-fragment-major layout [M/4][H][4] (fp32 kernel)
    {
        std::vector<float> r = host(C0r, (size_t)M * H), f((size_t)M * H);
        for (int m = 0; m < M; ++m) for (int n = 0; n < H; ++n) f[((size_t)(m / 4) * H + n) * 4 + (m & 3)] = r[(size_t)m * H + n];
        CK(hipMalloc(&C0, (size_t)M * H * 4)); CK(hipMemcpy(C0, f.data(), (size_t)M * H * 4, hipMemcpyHostToDevice));
        // the split kernel's layout (GF_C_FRAGT): [M/16][H/16][q = (n % 16) / 4][c = m % 16][n % 4]
        for (int m = 0; m < M; ++m) for (int n = 0; n < H; ++n) f[(((size_t)(m / 16) * (H / 16) + n / 16) * 64 + ((n % 16) / 4) * 16 + m % 16) * 4 + n % 4] = r[(size_t)m * H + n];
        CK(hipMemcpy(C0r, f.data(), (size_t)M * H * 4, hipMemcpyHostToDevice));
    }
    float* W = dev((size_t)H * H, 3, 0.08f);
    float* W0 = dev((size_t)48 * H, 4, 0.2f);        // layer-0 kernel rows (obs | act | t | pad), ld = H
    float* W4 = dev((size_t)H * ap, 5, 0.1f);
    float* bias = dev(H, 6, 0.1f);
    float* eb = dev(ap, 7, 0.1f);
    float* ea_in = dev((size_t)M * 48, 8, 1.0f);
    float* evp_in = dev((size_t)(H / 32) * M * ap, 9, 0.1f);
    float *Wf[2], *W0f[2], *W4f[2], *Cout[2], *ea_out[2], *evp_out[2];
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&Wf[v], (size_t)H * H * 4)); CK(hipMalloc(&W0f[v], (size_t)16 * H * 4)); CK(hipMalloc(&W4f[v], (size_t)H * ap * 4));
        CK(hipMalloc(&Cout[v], (size_t)M * H * 4)); CK(hipMalloc(&ea_out[v], (size_t)M * ap * 4)); CK(hipMalloc(&evp_out[v], (size_t)(H / 32) * M * ap * 4));
        std::vector<WfragTask> t;
        int tile = 0;
        auto add = [&](const float* src, float* dst, int K, int N, int ld, int kvalid, int mode) {
            t.push_back(WfragTask{src, dst, K, N, ld, kvalid, tile, v ? mode : 0});
            tile += ((K / 4) * N + FQL_THREADS - 1) / FQL_THREADS;
        };
        add(W, Wf[v], H, H, H, H, 1);
        add(W0 + (size_t)od * H, W0f[v], 16, H, H, ad + 1, 2);
        add(W4, W4f[v], H, ap, ap, H, 1);
        WfragTask* dt; CK(hipMalloc(&dt, t.size() * sizeof(WfragTask)));
        CK(hipMemcpy(dt, t.data(), t.size() * sizeof(WfragTask), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fql_wfrag_kernel, dim3(tile), dim3(FQL_THREADS), 0, s, (const WfragTask*)dt, (int)t.size(), -1);
        CK(hipStreamSynchronize(s));
    }
    for (int variant = 0; variant < 3; ++variant) {
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(Cout[v], 0, (size_t)M * H * 4)); CK(hipMemset(ea_out[v], 0, (size_t)M * ap * 4)); CK(hipMemset(evp_out[v], 0, (size_t)(H / 32) * M * ap * 4));
            ChainArgs a{};
            a.A = variant == 0 ? (v ? C0r : C0) : (v ? As : A); a.Wf = Wf[v]; a.bias = bias; a.C = Cout[v];
            a.ea_in = ea_in; a.ea_out = ea_out[v]; a.W0f = W0f[v]; a.evp_in = evp_in; a.eb = eb;
            a.W4f = W4f[v]; a.evp_out = evp_out[v];
            a.M = M; a.ad = ad; a.ap = ap; a.ea_ld = 48; a.inv_steps = 0.1f; a.t_s = 0.3f; a.variant = variant; a.tl = -1; a.prio = 0; a.stamps = nullptr;
            if (v == 0) { if (variant == 0) hipLaunchKernelGGL((fql_chain_kernel<512, 0>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); else if (variant == 1) hipLaunchKernelGGL((fql_chain_kernel<512, 1>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); else hipLaunchKernelGGL((fql_chain_kernel<512, 2>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); }
            else if (variant == 0) hipLaunchKernelGGL((fql_chain_split_kernel<512, 0>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            else if (variant == 1) hipLaunchKernelGGL((fql_chain_split_kernel<512, 1>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            else hipLaunchKernelGGL((fql_chain_split_kernel<512, 2>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            // timing: 96 launches per graph
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < 96; ++i) {
                if (v == 0) { if (variant == 0) hipLaunchKernelGGL((fql_chain_kernel<512, 0>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); else if (variant == 1) hipLaunchKernelGGL((fql_chain_kernel<512, 1>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); else hipLaunchKernelGGL((fql_chain_kernel<512, 2>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a); }
                else if (variant == 0) hipLaunchKernelGGL((fql_chain_split_kernel<512, 0>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            else if (variant == 1) hipLaunchKernelGGL((fql_chain_split_kernel<512, 1>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            else hipLaunchKernelGGL((fql_chain_split_kernel<512, 2>), dim3((M / 16) * 16), dim3(FQL_CHAIN_THREADS), FQL_CHAIN_LDS_BYTES(512), s, a);
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / (30.0 * 96);
            printf("variant %c %s: %.2f us per launch\n", "ABC"[variant], v ? "split" : "fp32 ", us);
        }
        double r;
        if (variant != 2) { const double d = maxdiff(host(Cout[0], (size_t)M * H), join_planes(host(Cout[1], (size_t)M * H), M, H), &r); printf("  C      max|fp32 - split| = %.3e (max|fp32| %.3f)\n", d, r); }
        if (variant == 0) { const double d = maxdiff(host(ea_out[0], (size_t)M * ap), host(ea_out[1], (size_t)M * ap), &r); printf("  ea_out max diff = %.3e (max %.3f)\n", d, r); }
        if (variant == 2) { const double d = maxdiff(host(evp_out[0], (size_t)(H / 32) * M * ap), host(evp_out[1], (size_t)(H / 32) * M * ap), &r); printf("  evp    max diff = %.3e (max %.3f)\n", d, r); }
    }
    return 0;
}
