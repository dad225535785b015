// Experiment (round 3): price of one dependent [rows x 512] x [512 x 512] + GELU layer when the batch is partitioned over the 8 XCDs
// and the 32 CUs of an XCD exchange activations through THEIR L2 only (plain stores, sc1 loads, one arrival counter per XCD) inside ONE
// persistent launch - no kernel boundary, no cross-XCD traffic, weights resident in LDS.  Variants by argv:
//   xcd_phase <phases> <mode> <rt>      mode bit0: sc1 (write-through) stores, bit1: skip MFMAs, bit2: skip waits (results invalid), bit3: group = blockIdx % 8
//                                       rt = 16-row tiles per XCD (2 = batch 256, 8 = batch 1024)
// Groups are formed from HW_REG_XCC_ID (the XCD a workgroup REALLY runs on), members by a per-XCD ticket: results do not depend on
// placement as long as every XCD receives 32 workgroups (census printed).  Output is checked against a host fp64 chain.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

struct Args {
    float* act[2];        // [8 groups][RT][32 j][64 lanes][4]
    const float* Wf;      // [3 layers][32 members][32 j][64][4]
    const float* bias;    // [3][512]
    unsigned* cnt;        // [8][32] (128 B apart)
    unsigned* ticket;     // [8][32]
    unsigned* err;
    unsigned long long* stamps;  // [256][phases][4]
    unsigned* census;     // [256] xcc id per block
    int phases, mode, rt;
};

__device__ __forceinline__ float gelu_f(float x) {
    const float x2 = x * x;
    const float e = fminf(__builtin_amdgcn_exp2f(x * fmaf(-0.1029432427f, x2, -2.302208198f)), 1e30f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// mode bit 4: the four waves synchronise through a counter in LDS (the two-team kernel's xsync) instead of s_barrier
#define SYNC() do { if (a.mode & 16) { bt += 4u; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bt) { if (a.mode & 32) __builtin_amdgcn_s_sleep(1); } asm volatile("" ::: "memory"); } else __syncthreads(); } while (0)
template <int RT>
__global__ __launch_bounds__(256) void xcd_chain(Args a) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];   // [3][2048] weights, [4][RT][64] reduce
    __shared__ unsigned s_member;
    __shared__ unsigned s_bar;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bt = 0u;
    __attribute__((address_space(3))) unsigned* bar = (__attribute__((address_space(3))) unsigned*)&s_bar;
    if (tid == 0) s_bar = 0u;
    const unsigned xcc = (a.mode & 8) ? (blockIdx.x & 7) : __builtin_amdgcn_s_getreg(6164);   // HW_REG_XCC_ID[3:0]
    if (tid == 0) {
        s_member = __hip_atomic_fetch_add((GAS unsigned*)(a.ticket + 32 * xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.census[blockIdx.x] = __builtin_amdgcn_s_getreg(6164);
    }
    __syncthreads();
    const unsigned member = s_member;
    if (member >= 32) { if (tid == 0) *a.err = 1; return; }
    // weights of the three layers: this member's 16 output columns, fragment-major, LDS resident
    for (int l = 0; l < 3; ++l)
        for (int i = tid; i < 2048; i += 256) lds[l * 2048 + i] = *(const GAS f32x4*)(a.Wf + ((size_t)(l * 32 + member) * 2048 + i) * 4);
    f32x4* red = lds + 3 * 2048;
    __syncthreads();
    GAS unsigned* cnt = (GAS unsigned*)(a.cnt + 32 * xcc);
    const size_t gstride = (size_t)RT * 32 * 64 * 4;   // floats per group
    unsigned long long* st = a.stamps + (size_t)blockIdx.x * a.phases * 4;
    for (int p = 0; p < a.phases; ++p) {
        if (p > 0 && !(a.mode & 4)) {
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u * p) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000u) { *a.err = 2; break; }
                }
            }
            SYNC();
        }
        if (tid == 0) st[p * 4 + 0] = __builtin_amdgcn_s_memtime();
        const float* src = a.act[p & 1] + xcc * gstride;
        float* dst = a.act[(p + 1) & 1] + xcc * gstride;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(gstride * 4), 0x00020000);
        const f32x4* wl = lds + (p % 3) * 2048;
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0, 0, 0, 0};
        // K split over the four waves: wave w takes j = 8 w .. 8 w + 7 (128 of the 512 inputs)
#pragma unroll
        for (int rt0 = 0; rt0 < RT; rt0 += 2) {
            f32x4 av[2][8];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    av[r2][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (((rt0 + r2) * 32 + 8 * w + jj) * 64 + lane) * 16, 0, 16));
            if (!(a.mode & 2)) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const f32x4 b = wl[(8 * w + jj) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[rt0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t], av[0][jj][t], acc[rt0], 0, 0, 0);
                        acc[rt0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t], av[1][jj][t], acc[rt0 + 1], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { acc[rt0] += av[0][jj]; acc[rt0 + 1] += av[1][jj]; }
            }
        }
        if (tid == 0) st[p * 4 + 1] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) red[(w * RT + rt) * 64 + lane] = acc[rt];
        SYNC();
        for (int o = tid; o < RT * 64; o += 256) {   // o = rt * 64 + lane'
            f32x4 s = red[o];
#pragma unroll
            for (int k = 1; k < 4; ++k) s += red[k * RT * 64 + o];
            const int q = (o & 63) >> 4;
            const f32x4 bv = *(const GAS f32x4*)(a.bias + (p % 3) * 512 + member * 16 + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = gelu_f(s[t] + bv[t]);
            const int rt = o >> 6;
            float* d = dst + (((size_t)rt * 32 + member) * 64 + (o & 63)) * 4;
            if (a.mode & 1) {
                __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)(gstride * 4), 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), rd, (int)((((size_t)rt * 32 + member) * 64 + (o & 63)) * 16), 0, 16);
            } else
                *(GAS f32x4*)d = s;
        }
        if (tid == 0) st[p * 4 + 2] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SYNC();
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st[p * 4 + 3] = __builtin_amdgcn_s_memtime();
        }
    }
}

static size_t fidx(int RT, int row, int col) {   // inside one group
    const int rt = row / 16, r = row % 16, j = col / 16, q = (col % 16) / 4, t = col % 4;
    return (((size_t)rt * 32 + j) * 64 + q * 16 + r) * 4 + t;
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 40, mode = argc > 2 ? atoi(argv[2]) : 0, RT = argc > 3 ? atoi(argv[3]) : 2;
    const int rows = 16 * RT, B = 8 * rows;
    std::vector<float> W(3 * 512 * 512), bias(3 * 512), X((size_t)B * 512);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : W) v = rnd() * 0.09f;
    for (auto& v : bias) v = rnd() * 0.1f;
    for (auto& v : X) v = rnd();
    std::vector<float> Wf((size_t)3 * 32 * 2048 * 4), A0((size_t)8 * RT * 32 * 64 * 4);
    for (int l = 0; l < 3; ++l)
        for (int c = 0; c < 32; ++c)
            for (int j = 0; j < 32; ++j)
                for (int ln = 0; ln < 64; ++ln)
                    for (int t = 0; t < 4; ++t)
                        Wf[(((size_t)(l * 32 + c) * 32 + j) * 64 + ln) * 4 + t] = W[(size_t)l * 512 * 512 + (size_t)(16 * j + 4 * (ln / 16) + t) * 512 + 16 * c + ln % 16];
    const size_t gs = (size_t)RT * 32 * 64 * 4;
    for (int g = 0; g < 8; ++g)
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < 512; ++c) A0[g * gs + fidx(RT, r, c)] = X[(size_t)(g * rows + r) * 512 + c];
    Args a{};
    float* dW; float* db;
    CK(hipMalloc(&a.act[0], A0.size() * 4)); CK(hipMalloc(&a.act[1], A0.size() * 4));
    CK(hipMalloc(&dW, Wf.size() * 4)); CK(hipMalloc(&db, bias.size() * 4));
    CK(hipMalloc(&a.cnt, 8 * 32 * 4)); CK(hipMalloc(&a.ticket, 8 * 32 * 4)); CK(hipMalloc(&a.err, 64)); CK(hipMalloc(&a.census, 256 * 4));
    CK(hipMalloc(&a.stamps, (size_t)256 * phases * 4 * 8));
    CK(hipMemcpy(dW, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    a.Wf = dW; a.bias = db; a.phases = phases; a.mode = mode; a.rt = RT;
    const size_t ldsb = (size_t)(3 * 2048 + 4 * RT * 64) * 16;
    auto kern = RT == 2 ? xcd_chain<2> : RT == 4 ? xcd_chain<4> : xcd_chain<8>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e9, sum = 0; const int reps = 20;
    for (int it = 0; it < reps + 3; ++it) {
        CK(hipMemcpyAsync(a.act[0], A0.data(), A0.size() * 4, hipMemcpyHostToDevice, s));
        CK(hipMemsetAsync(a.cnt, 0, 8 * 32 * 4, s)); CK(hipMemsetAsync(a.ticket, 0, 8 * 32 * 4, s)); CK(hipMemsetAsync(a.err, 0, 64, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), ldsb, s, a);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) { best = std::min(best, (double)ms * 1e3); sum += ms * 1e3; }
    }
    unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> cen(256); CK(hipMemcpy(cen.data(), a.census, 1024, hipMemcpyDeviceToHost));
    int per[8] = {0}, rr = 0; for (int b = 0; b < 256; ++b) { per[cen[b] & 7]++; rr += (cen[b] & 7) == ((cen[0] + b) & 7); }
    printf("phases=%d mode=%d RT=%d (batch %d): kernel %.1f us best / %.1f mean = %.2f us per phase; err=%u\n", phases, mode, RT, B, best, sum / reps, best / phases, err);
    printf("  census: per-XCD workgroups %d %d %d %d %d %d %d %d; round-robin from block 0's XCD (%u): %d / 256\n", per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], cen[0], rr);
    // in-kernel stamps (shader clocks), averaged over workgroups and phases >= 1
    std::vector<unsigned long long> st((size_t)256 * phases * 4);
    CK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double d[4] = {0}; size_t n = 0;
    for (int b = 0; b < 256; ++b)
        for (int p = 1; p < phases; ++p) {
            const unsigned long long* q = &st[((size_t)b * phases + p) * 4];
            d[0] += (double)(q[0] - q[-1]);   // previous arrive -> wait over
            d[1] += (double)(q[1] - q[0]);    // loads + MFMA
            d[2] += (double)(q[2] - q[1]);    // reduce + epilogue + stores issued
            d[3] += (double)(q[3] - q[2]);    // drain + barrier + arrive
            ++n;
        }
    const double tot = d[0] + d[1] + d[2] + d[3];
    printf("  stamps (share of a phase): wait %.1f%% | loads+MFMA %.1f%% | reduce+epilogue %.1f%% | drain+arrive %.1f%%  (%.0f clocks per phase)\n", 100 * d[0] / tot, 100 * d[1] / tot,
           100 * d[2] / tot, 100 * d[3] / tot, tot / n);
    // check against the host chain (fp64)
    if (!(mode & 6)) {
        std::vector<float> out(A0.size());
        CK(hipMemcpy(out.data(), a.act[phases & 1], out.size() * 4, hipMemcpyDeviceToHost));
        const int chk_rows = std::min(B, 64);
        std::vector<double> cur((size_t)chk_rows * 512), nxt((size_t)chk_rows * 512);
        std::vector<int> rsel(chk_rows);
        for (int i = 0; i < chk_rows; ++i) { rsel[i] = (int)((long long)i * B / chk_rows); for (int c = 0; c < 512; ++c) cur[(size_t)i * 512 + c] = X[(size_t)rsel[i] * 512 + c]; }
        for (int p = 0; p < phases; ++p) {
            const float* wl = &W[(size_t)(p % 3) * 512 * 512];
            for (int i = 0; i < chk_rows; ++i)
                for (int n2 = 0; n2 < 512; ++n2) {
                    double s2 = bias[(p % 3) * 512 + n2];
                    for (int k = 0; k < 512; ++k) s2 += cur[(size_t)i * 512 + k] * wl[(size_t)k * 512 + n2];
                    const double u = 0.7978845608028654 * (s2 + 0.044715 * s2 * s2 * s2);
                    nxt[(size_t)i * 512 + n2] = 0.5 * s2 * (1.0 + tanh(u));
                }
            cur.swap(nxt);
        }
        double maxe = 0, maxv = 0;
        for (int i = 0; i < chk_rows; ++i)
            for (int c = 0; c < 512; ++c) {
                const int g = rsel[i] / rows, r = rsel[i] % rows;
                const double v = out[g * gs + fidx(RT, r, c)];
                maxe = std::max(maxe, std::fabs(v - cur[(size_t)i * 512 + c])); maxv = std::max(maxv, std::fabs(cur[(size_t)i * 512 + c]));
            }
        printf("  check vs host fp64 chain (%d rows): max |err| %.3g, max |value| %.3g  %s\n", chk_rows, maxe, maxv, maxe < 1e-4 * std::max(1.0, maxv) ? "OK" : "MISMATCH");
    }
    return 0;
}
