// Euler chain of the BC flow (agents/fql.py:155-171) on gfx950: the three launches of one Euler step as ONE
// specialised kernel family.
//
// The chain is flow_steps x (4 hidden layers + head) strictly sequential [B x H] x [H x H] products: 30 launches
// per update that are latency-bound, not throughput-bound (0.134 GFLOP each = 0.85 us of the chip's fp32 matrix
// rate).  What a launch costs is its serial per-wave work, so this kernel halves all of it relative to the generic
// 16-row kernel (fql_gemm16_kernel):
//   * 512-thread workgroups on a 16 x 32 output tile: 8 waves = 2 column tiles x 4 K-quarters, 32 MFMAs per wave
//     instead of 64, two waves per SIMD;
//   * weights come from a FRAGMENT-MAJOR copy Wf[k/4][n][4] written once per update behind Adam
//     (fql_wfrag_kernel): lane (c, q) of k-group g needs W[16g + 4q + s][n0 + c], s = 0..3, which is ONE 16-byte load
//     there (four 4-byte loads of four different rows in the row-major arena).  The whole K-quarter of a wave is 8 such
//     loads, all issued before anything else - no chunk loop;
//   * the task is the kernel argument (no task-table hop through HBM before the first load can issue);
//   * the K-quarter partials meet in LDS and the 8 waves share the epilogue (one output element per lane).
// Variants (agents/fql.py:166-169):
//   A  fold the previous step's head partials -> a_s ; layer 0 as a rank-(act + 1) update of the loop-invariant
//      C0 = obs W0 + b0 ; GELU ; layer 1
//   B  a middle hidden layer
//   C  the last hidden layer + the action head as per-column-tile partials (folded by the next step's variant A, or by
//      fql_euler_finish_kernel after the last step) - fixed summation order, deterministic.
// Numerics: fp32 MFMA (v_mfma_f32_16x16x4_f32) = exact fp32 fma chains; only the summation ORDER over k differs from
// the generic kernel (K quarters instead of halves).
#pragma once
#include "fql_kernels.h"

#define FQL_CHAIN_THREADS 512
#ifndef FQL_CHAIN_WAVES
#define FQL_CHAIN_WAVES 4   // min waves per SIMD the register allocation must allow (4: <= 128 VGPRs, the kernel needs 102)
#endif

struct ChainArgs {
    const float* A;       // B, C: input activations [M, H] row-major.  A: C0 in C-fragment-major layout [M/4][H][4]
    const float* Wf;      // fragment-major kernel of this launch's H x H layer: [H/4][H][4]
    const float* bias;    // [H]
    float* C;             // A, B: output activations [M, H] row-major
    union {
        struct {   // variants A / C
            const float* ea_in;   // [M, ea_ld] actions a_{s-1} (step 0: the noise z)
            float* ea_out;        // [M, ap] a_s as used by this step (written by column tile 0) or null
            const float* W0f;     // fragment-major W0 rows of (action block, t, zero padding): [4][H][4]
            const float* evp_in;  // [H/32][M][ap] head partials of the previous step, or null (step 0)
            const float* eb;      // [ap] head bias
            const float* W4f;     // fragment-major head kernel [H/4][ap][4] (variant C)
            float* evp_out;       // [H/32][M][ap] (variant C)
        };
        struct {   // variant E (below)
            const float* Gv;      // [M, H] GELU(z) of the layer whose LayerNorm is differentiated = what the LayerNorm normalised
            const float* stats;   // [M, 2] mean, rstd
            const float* gamma;   // [H] LayerNorm scale
            const float* dq;      // scalar head: dY[m][k] = dq[m ldq] wq[k ldw] when A is null, else unused
            const float* wq;
            int ldq, ldw;
            int width;            // real (unpadded) layer width: LayerNorm statistics run over k < width
            int ncol, ldc;        // output columns (multiple of 32; H for the hidden layers, the padded input width for layer 0) and row stride of C
        };
    };
    int M, ad, ap, ea_ld;
    float inv_steps, t_s;
    int variant;          // 0 = A, 1 = B, 2 = C, 3 = D (below)
    // variant D: an input-gradient layer of a plain (un-normalised) MLP as a chain launch, dX = (dZ W^T) * GELU'(z_prev)
    // (utils/flax_utils.py:137 through utils/networks.py:53-58): A = dZ [M, H]; Wf = the ROW-MAJOR kernel W[j][n] - a lane's four
    // contraction values n = 16 g + 4 q + s of output column j are 16 contiguous bytes there, no copy needed; no bias, no GELU
    const float* Zprev;   // [M, H] stored GELU'(z) of the previous layer
    // variant E: one level of the input-gradient chain through a LayerNorm'd MLP (the critic's Q-gradient path, agents/fql.py:69-79 through
    // utils/networks.py:53-58) in ONE launch instead of a LayerNorm-backward launch + a dgrad launch: the prologue rebuilds
    //   dZ = rstd (d - mean(d) - xhat mean(d xhat)) GELU'(z),  d = dY gamma,  xhat = (GELU(z) - mean) rstd
    // for the workgroup's 16 rows in LDS (whole rows are there, so the two row statistics need no second pass), then dX = dZ W^T with the
    // row-major kernel as in variant D; no epilogue (the next level's prologue applies its own LayerNorm backward).  A = dY [M, H] or null
    // (scalar head: dY = dq (x) wq), Zprev = GELU'(z), C = dX [M, ldc].  fp32 operands in both precisions.
    int xg;               // XCD-aware tile order (xcd_tile) with xg row groups; 0 = row-major
    int hw;               // variant E: hidden width of THIS launch's layers (the critic's; the other variants use the BC flow's)
    int tl;               // timeline id (diagnostics build)
    int prio;             // s_setprio level of the chain's waves (they are latency-critical and light: 0.85 us of MFMA per launch)
    unsigned long long* stamps;   // diagnostics build only (FQL_STAMPS): [grid][8] wall-clock stamps
};

// dst[(k >> 2) * N + n][k & 3] = src[k * ld + n]: the layout the chain kernel's B fragments are one dwordx4 in
struct WfragTask {
    const float* src;
    float* dst;
    int K, N, ld;     // K multiple of 4
    int kvalid;       // rows [kvalid, K) are written as zeros (the rank-update block may run past the layer's padded input rows)
    int tile0;        // first workgroup of this task
    int mode;         // 0: fp32 Wf[k/4][n][4].  precision = 2 (fql_chain_split_kernel): 1 = split 16x16x32 operands
                      // dst[k/32][hi, lo][n][q][4 words], lane (n, q) owns k = 32 j + 8 q .. + 7 (K multiple of 32; a wave's 16-byte
                      // loads of one plane are 1 KB contiguous);
                      // 2 = split 16x16x16 operands dst[(k/4) N + n][hi 2 words | lo 2 words] (the 16-row rank-update block)
};
__global__ __launch_bounds__(FQL_THREADS) void fql_wfrag_kernel(const WfragTask* __restrict__ tasks, int ntasks, int tl) {
    tl_enter(tl);
    const WfragTask& T = tasks[find_task(tasks, ntasks, blockIdx.x)];
    const int e = (blockIdx.x - T.tile0) * FQL_THREADS + threadIdx.x;   // one (k4, n) per thread
    if (T.mode == 1) {   // one (j, q, n) per thread: 8 k values -> 32 bytes
        const int n = e % T.N, r = e / T.N;
        const int q = r & 3, j = r >> 2;
        if (j >= (T.K >> 5)) return;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int k = 32 * j + 8 * q + i; v[i] = (k < T.kvalid) ? ldg(T.src + (size_t)k * T.ld + n) : 0.f; }
        u32x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { unsigned h, l; bsplit2(v[2 * i], v[2 * i + 1], h, l); hi[i] = h; lo[i] = l; }
        unsigned* d = reinterpret_cast<unsigned*>(T.dst) + (((size_t)(2 * j) * T.N + n) * 4 + q) * 4;
        *(FQL_GAS u32x4*)d = hi;
        *(FQL_GAS u32x4*)(d + (size_t)T.N * 16) = lo;
        return;
    }
    if (T.mode == 2) {
        const int n = e % T.N, k4 = e / T.N;
        if (k4 >= (T.K >> 2)) return;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (4 * k4 + i < T.kvalid) ? ldg(T.src + (size_t)(4 * k4 + i) * T.ld + n) : 0.f;
        unsigned h0, l0, h1, l1;
        bsplit2(v[0], v[1], h0, l0);
        bsplit2(v[2], v[3], h1, l1);
        *(FQL_GAS u32x4*)(reinterpret_cast<unsigned*>(T.dst) + ((size_t)k4 * T.N + n) * 4) = u32x4{h0, h1, l0, l1};
        return;
    }
    const int n = e % T.N, k4 = e / T.N;
    if (k4 >= (T.K >> 2)) return;
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (4 * k4 + s < T.kvalid) ? ldg(T.src + (size_t)(4 * k4 + s) * T.ld + n) : 0.f;
    stg4(T.dst + ((size_t)k4 * T.N + n) * 4, v);
}

template <int H, int V>   // V: the variant as a compile-time constant (each variant gets the register allocation of its own code)
__device__ __forceinline__ void chain_body(const ChainArgs& P) {
    static_assert(H % 128 == 0 && H <= 1024, "hidden width must be a multiple of 128");
    constexpr int S = H + 4;          // LDS row stride of the A tile (floats)
    constexpr int GQ = H / 64;        // k-groups (of 16) per K-quarter
    constexpr int NA = H / 128;       // float4 loads per thread that cover the 16 x H A tile
    constexpr int CT = H / 128;       // layer-0 column tiles per wave (variant A)
    constexpr int NT = H / 32;        // column tiles = head partials per row tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* red = lds + 16 * S;        // [4 K-quarters][2 column tiles][64 lanes] float4
    float* ea = red + 4 * 2 * 64 * 4; // [16][32] actions of this step (variant A)
    float* hs = ea + 512;             // [16][36] GELU tile feeding the head partial (variant C)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int nt = wave & 1, kp = wave >> 1;
    constexpr int variant = V;
    const int ntc = variant == 4 ? P.ncol / 32 : NT;
    int tm = blockIdx.x / ntc, tn = blockIdx.x - tm * ntc;
    if (P.xg) xcd_tile((int)blockIdx.x, P.M >> 4, ntc, P.xg, tm, tn);
    const int row0 = tm * 16, n0 = tn * 32 + nt * 16;
    tl_enter(P.tl);
    if (P.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (P.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (P.prio == 1) __builtin_amdgcn_s_setprio(1);
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define CSTAMP() do { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSTAMP() do {} while (0)
#endif
    CSTAMP();
    // ---- every load of the launch that does not depend on another workgroup's data of THIS launch goes out first
    f32x4 bf[GQ];
    {
        if (variant == 3 || variant == 4) {   // row-major kernel read transposed: output column j = n0 + c, contraction n = 16 (kp GQ + g) + 4 q + s
            const float* wb = P.Wf + (size_t)(n0 + c) * H + 16 * kp * GQ + 4 * q;
#pragma unroll
            for (int g = 0; g < GQ; ++g) bf[g] = ldg4(wb + 16 * g);
        } else {
            const float* wb = P.Wf + ((size_t)(4 * kp * GQ + q) * H + n0 + c) * 4;
#pragma unroll
            for (int g = 0; g < GQ; ++g) bf[g] = ldg4(wb + (size_t)g * 16 * H);   // k-group kp GQ + g: rows 4 (kp GQ + g) + q of Wf
        }
    }
    // epilogue operand of this lane's output element (row 4q + kp, column n0 + c): the bias, or GELU'(z_prev) for variant D
    const float bias = variant == 4 ? 0.f : variant == 3 ? ldg(P.Zprev + (size_t)(row0 + 4 * q + kp) * H + n0 + c) : ldg(P.bias + n0 + c);
    f32x4 bw4[2];
    if (variant == 2 && wave == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) bw4[g] = ldg4(P.W4f + ((size_t)(tn * 8 + 4 * g + q) * P.ap + c) * 4);
    }
    if (variant == 0) {
        // layer 0: C0 (loop invariant) + [a_s | t_s | 0] (16 x 16) times the 16 rows of W0 that start at the action block
        f32x4 cacc[CT], wf[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int ct = wave + 8 * t;
            cacc[t] = ldg4(P.A + ((size_t)((row0 >> 2) + q) * H + 16 * ct + c) * 4);   // C layout: rows 4q + i, column c
            wf[t] = ldg4(P.W0f + ((size_t)q * H + 16 * ct + c) * 4);                   // k = 4q + s
        }
        {   // a_s = a_{s-1} + (sum of head partials + head bias) / flow_steps: two threads per element, fixed order
            const int e = tid >> 1, half = tid & 1;
            const int r = e >> 4, j = e & 15;
            float a = 0.f;
            if (j < P.ad) {
                a = ldg(P.ea_in + (size_t)(row0 + r) * P.ea_ld + j);
                if (P.evp_in) {
                    float pv[NT / 2];
                    const float* pp = P.evp_in + ((size_t)(half * (NT / 2)) * P.M + row0 + r) * P.ap + j;
#pragma unroll
                    for (int tp = 0; tp < NT / 2; ++tp) pv[tp] = ldg(pp + (size_t)tp * P.M * P.ap);
                    float sum = 0.f;
#pragma unroll
                    for (int tp = 0; tp < NT / 2; ++tp) sum += pv[tp];
                    const float other = __shfl_xor(sum, 1);
                    const float tot = half ? other + sum : sum + other;   // (partials 0..NT/2-1) + (NT/2..NT-1) on both lanes
                    a += (tot + ldg(P.eb + j)) * P.inv_steps;
                }
                if (tn == 0 && half == 0 && P.ea_out) stg(P.ea_out + (size_t)(row0 + r) * P.ap + j, a);
            } else if (j == P.ad) {
                a = P.t_s;
            }
            if (half == 0) ea[r * 32 + j] = a;
        }
        CSTAMP();
        __syncthreads();
        {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&ea[c * 32 + 4 * q]);   // A'[row c][k = 4q + s]
#pragma unroll
            for (int t = 0; t < CT; ++t) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) cacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], wf[t][s4], cacc[t], 0, 0, 0);
                const int ct = wave + 8 * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) lds[(4 * q + i) * S + 16 * ct + c] = gelu_f(cacc[t][i]);
            }
        }
    } else if (variant == 4) {
        // LayerNorm backward of the workgroup's 16 rows into the A tile: wave w owns rows 2w, 2w + 1; a lane owns the float4 columns
        // 4 (lane + 64 i), i < H / 256, of both rows.  Every load first, then the two row sums per row by wave shuffles.
        constexpr int NV = H / 256;
        f32x4 dy[2][NV], zz[2][NV], gq[2][NV], gm[NV];
        float mean[2], rstd[2];
#pragma unroll
        for (int i = 0; i < NV; ++i) gm[i] = ldg4(P.gamma + 4 * (lane + 64 * i));
        float wqv[NV][4];
        if (!P.A) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) wqv[i][e] = ldg(P.wq + (size_t)(4 * (lane + 64 * i) + e) * P.ldw);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t row = (size_t)(row0 + 2 * wave + r);
            mean[r] = ldg(P.stats + 2 * row); rstd[r] = ldg(P.stats + 2 * row + 1);
            const float dqr = P.A ? 0.f : ldg(P.dq + row * P.ldq);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k4 = 4 * (lane + 64 * i);
                zz[r][i] = ldg4(P.Zprev + row * H + k4);
                gq[r][i] = ldg4(P.Gv + row * H + k4);
                if (P.A) dy[r][i] = ldg4(P.A + row * H + k4);
                else dy[r][i] = f32x4{dqr * wqv[i][0], dqr * wqv[i][1], dqr * wqv[i][2], dqr * wqv[i][3]};
            }
        }
        const float inv = 1.0f / (float)P.width;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = dy[r][i][e] * gm[i][e], xh = (gq[r][i][e] - mean[r]) * rstd[r];
                    dy[r][i][e] = d; gq[r][i][e] = xh;
                    if (4 * (lane + 64 * i) + e < P.width) { s1 += d; s2 += d * xh; }
                }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 * inv, m2 = s2 * inv;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = (4 * (lane + 64 * i) + e < P.width) ? rstd[r] * (dy[r][i][e] - m1 - gq[r][i][e] * m2) * zz[r][i][e] : 0.f;
                *reinterpret_cast<f32x4*>(&lds[(2 * wave + r) * S + 4 * (lane + 64 * i)]) = o;
            }
        }
    } else {
        f32x4 av[NA];
        const float* Ag = P.A + (size_t)row0 * H;
#pragma unroll
        for (int i = 0; i < NA; ++i) av[i] = ldg4(Ag + (size_t)(tid + i * FQL_CHAIN_THREADS) * 4);   // 16 rows of H floats are contiguous
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * FQL_CHAIN_THREADS;
            const int r = f / (H / 4), kk = f - r * (H / 4);
            *reinterpret_cast<f32x4*>(&lds[r * S + 4 * kk]) = av[i];
        }
        CSTAMP();
    }
    __syncthreads();
    CSTAMP();
    // ---- this wave's K-quarter: GQ k-groups x 4 MFMAs on two accumulator chains
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    {
        const float* arow = lds + c * S + 4 * q + 16 * kp * GQ;
#pragma unroll
        for (int g = 0; g < GQ; g += 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * g);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(arow + 16 * g + 16);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], bf[g][s4], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s4], bf[g + 1][s4], acc2, 0, 0, 0);
            }
        }
        acc += acc2;
    }
    *reinterpret_cast<f32x4*>(&red[((kp * 2 + nt) * 64 + lane) * 4]) = acc;
    CSTAMP();
    __syncthreads();
    // ---- epilogue shared by the 8 waves: wave (nt, kp) finishes row 4q + kp, column n0 + c of its column tile
    float v = (variant == 3 || variant == 4) ? 0.f : bias;
#pragma unroll
    for (int p = 0; p < 4; ++p) v += red[((p * 2 + nt) * 64 + lane) * 4 + kp];
    v = variant == 4 ? v : variant == 3 ? v * bias : gelu_f(v);
    const int row = row0 + 4 * q + kp;
    if (variant == 4) {
        stg(P.C + (size_t)row * P.ldc + n0 + c, v);
    } else if (variant != 2) {
        stg(P.C + (size_t)row * H + n0 + c, v);
    } else {
        hs[(4 * q + kp) * 36 + 16 * nt + c] = v;
        __syncthreads();
        if (wave == 0) {   // 16 x 32 GELU tile times this workgroup's 32 rows of the head kernel
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&hs[c * 36 + 16 * g + 4 * q]);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], bw4[g][s4], pa, 0, 0, 0);
            }
            if (c < P.ap) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stg(P.evp_out + ((size_t)tn * P.M + row0 + 4 * q + i) * P.ap + c, pa[i]);
            }
        }
    }
    tl_exit(P.tl);
#ifdef FQL_STAMPS
    CSTAMP();
    if (lane == 0 && wave == 0 && P.stamps) {
        unsigned long long* d = P.stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}
// dQ/da of the critic members (agents/fql.py:69-72 through utils/networks.py:53-58): the input gradient of the critic's first layer, of which a state agent
// needs nothing but the ACTION block - 16 columns of dX0 = dZ0 W0^T starting at `col0` (the observation part feeds nobody: no encoder behind it).  It is the
// last launch of the Q-gradient lane and what the critical lane's tail waits for.  The generic 16-row kernel computed all 64 input columns with 216 registers
// a lane (so it could not sit beside the side lanes' workgroups and took 11-15 us in the steady state); here: up to four tasks as the kernel argument, one
// workgroup per task and 16-row tile, the four waves split the contraction, every operand of a wave is eight 16-byte loads of one kernel row / one dZ row.
struct Dgrad0Args {
    GemmTask t[4];
    int ntasks, col0;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_dgrad0_kernel(const Dgrad0Args P) {
    __shared__ __attribute__((aligned(16))) float red[4 * 64 * 4];
    const int tid = threadIdx.x, kp = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int ntm = P.t[0].M >> 4;
    const int ti = (int)blockIdx.x / ntm, tm = (int)blockIdx.x - ti * ntm;
    const GemmTask& T = P.t[ti];
    const int row0 = tm * 16, kq = T.K >> 2;           // contraction columns per wave (a multiple of 16)
    const float* ap = T.A + (size_t)(row0 + c) * T.lda + kp * kq + 4 * q;          // dZ0[row c][k]
    const float* bp = T.B + (size_t)(P.col0 + c) * T.ldb + kp * kq + 4 * q;        // W0[input col0 + c][k] (row-major [in][out]: contiguous in k)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < kq; g0 += 128) {             // (K is a multiple of 512.)  8 k-groups of 16 per round: 16 loads in flight, then 32 MFMAs on two accumulator chains
        f32x4 a[8], b[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { a[g] = ldg4(ap + g0 + 16 * g); b[g] = ldg4(bp + g0 + 16 * g); }
#pragma unroll
        for (int g = 0; g < 8; g += 2)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s4], b[g][s4], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g + 1][s4], b[g + 1][s4], acc2, 0, 0, 0);
            }
    }
    acc += acc2;
    *reinterpret_cast<f32x4*>(&red[(kp * 64 + lane) * 4]) = acc;
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) v += red[(p * 64 + lane) * 4 + kp];      // wave kp finishes row 4q + kp, column c
    stg(T.C + (size_t)(row0 + 4 * q + kp) * T.ldc + P.col0 + c, v);
}

// The one-step actor's head dgrad (agents/fql.py:62-79 through utils/networks.py:53-58) as a launch of its own on the critical lane, between the Euler
// chain and the three tail dgrads: dX = (dA W_head^T) * GELU'(z_3) with dA = d(actor loss)/d(one-step actions) built in the prologue (GF_A_LOSSACT: distillation
// term against the Euler target + the clip-masked Q gradient of both critic members; with GF_A_EULFIN the target is finished here from the last step's head
// partials, same loads and summation order as fql_euler_finish_kernel).  The contraction is 16 wide - four MFMAs a wave - so the launch is its loads: the task
// arrives as the kernel argument (no table hop), one thread builds one dA element, a wave owns a 16 x 16 output tile.  It replaces the generic 16-row kernel's
// instantiation for this task (216 registers and 192 spilled scalars there: every other flag combination's state was live through it).
__global__ __launch_bounds__(FQL_THREADS) void fql_head_dgrad_kernel(const GemmTask T) {
    __shared__ __attribute__((aligned(16))) float da[16 * 20];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int ntn = T.N >> 6;
    const int tm = (int)blockIdx.x / ntn, tn = (int)blockIdx.x - tm * ntn;
    const int row0 = tm * 16, n0 = tn * 64 + 16 * wave;
    // what depends on nothing of this launch: the wave's slice of the head kernel (row-major [N][ap]: lane (c, q) reads W[n0 + c][4q .. 4q + 3]) and GELU'(z)
    const f32x4 b4 = ldg4(T.B + (size_t)(n0 + c) * T.ldb + 4 * q);
    float zp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) zp[i] = ldg(T.Zprev + (size_t)(row0 + 4 * q + i) * T.ldc + n0 + c);
    {
        const int r = tid >> 4, j = tid & 15;
        float g = 0.f;
        if (j < T.i2) {
            const float ar = ldg(T.ea_in + (size_t)(row0 + r) * T.i0 + j);
            float tg;
            if (T.flags & GF_A_EULFIN) {
                float pv[32];
#pragma unroll
                for (int tp = 0; tp < 32; ++tp) pv[tp] = ldg(T.aux2 + ((size_t)min(tp, T.ln_width - 1) * T.M + row0 + r) * T.i0 + j);
                float sum = 0.f;
#pragma unroll
                for (int tp = 0; tp < 32; ++tp) sum += (tp < T.ln_width) ? pv[tp] : 0.f;
                tg = clip1(ldg(T.aux + (size_t)(row0 + r) * T.i0 + j) + (sum + ldg(T.eb + j)) * T.f1);
                if (tn == 0) stg(const_cast<float*>(T.evp) + (size_t)(row0 + r) * T.i0 + j, tg);   // the actor-loss metrics read it
            } else tg = ldg(T.evp + (size_t)(row0 + r) * T.i0 + j);
            g = T.f0 * (ar - tg);
            if (ar > -1.0f && ar < 1.0f) {
                const size_t o = (size_t)(row0 + r) * T.e_ntp + T.i1 + j;
                g += ldg(T.ew + o) + ldg(T.ew4 + o);
            }
            if (tn == 0 && T.ea_out) stg(T.ea_out + (size_t)(row0 + r) * T.i0 + j, g);   // the head's weight gradient reads it
        }
        da[r * 20 + j] = g;
    }
    __syncthreads();
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&da[c * 20 + 4 * q]);   // dA[row c][k = 4q + s]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], b4[s4], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) stg(T.C + (size_t)(row0 + 4 * q + i) * T.ldc + n0 + c, acc[i] * zp[i]);
}

template <int H, int V>
__global__ __launch_bounds__(FQL_CHAIN_THREADS, FQL_CHAIN_WAVES) void fql_chain_kernel(const ChainArgs P) { chain_body<H, V>(P); }
// two independent tasks of one variant in one launch (blockIdx.y picks the task): the two ensemble members of the critic's Q-gradient chain
struct ChainPair { ChainArgs t[2]; };
template <int H, int V>
__global__ __launch_bounds__(FQL_CHAIN_THREADS, FQL_CHAIN_WAVES) void fql_chain_pair_kernel(const ChainPair P) { chain_body<H, V>(P.t[blockIdx.y]); }
// ------------------------------------------------------------------------------------------------
// precision = 2: the same three variants with split-bf16 operands (fql_kernels.h, top).  Weights arrive pre-split from
// fql_wfrag_kernel (modes 1 / 2: one 16-byte hi and one 16-byte lo load per lane and 32-deep MFMA step, the bytes of the fp32
// fragment copy); the 16 x H activation tile is split once by the thread that stages it into hi / lo planes [16][H/2 words] whose
// 4-word slots are XOR-swizzled with the row (rows are a multiple of 64 words apart: unswizzled, every row would sit on the same
// banks; with the swizzle both the b128 fragment reads and the b64 staging writes are conflict-free).  Per wave and K-quarter:
// 3 x H/128 MFMAs of 16 cycles instead of H/16 of 32.
// ------------------------------------------------------------------------------------------------
// value of the lane whose column index differs in bit 0 (DPP quad_perm [1, 0, 3, 2]: no LDS crossbar round trip)
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// Activations BETWEEN the launches of the split chain travel pre-split: a [M][H] tensor is a hi plane [M][H/2 words] followed by a
// lo plane of the same shape (the bytes of the fp32 tensor; the buffers are private to the chain).  The producer's epilogue splits
// each value once (lanes c, c ^ 1 pair up: the even one stores the hi word, the odd one the lo word); the consumer's staging is
// 16-byte copies into the swizzled LDS planes, no arithmetic.
// V: the variant as a compile-time constant, so each variant gets the register allocation of its own code: B and C need 62 / 61
// registers, A 89.  Why it matters: a side-lane workgroup holds 168 registers per SIMD and a chain workgroup two waves per SIMD, so
// next to TWO resident side workgroups a chain workgroup fits only with <= 88 registers per wave (2 x 88 + 2 x 168 = 512); at 104
// (one kernel for all variants) every chain launch waited ~2 us for CUs to drain to one side workgroup (steady-state timeline).
// Variant A is capped at 80 (6 waves per SIMD: two spilled values).
#ifndef FQL_CHAIN_SPLIT_WAVES_A
#define FQL_CHAIN_SPLIT_WAVES_A 4
#endif
template <int H, int V>
__global__ __launch_bounds__(FQL_CHAIN_THREADS, (V == 0 ? FQL_CHAIN_SPLIT_WAVES_A : 4)) void fql_chain_split_kernel(const ChainArgs P) {
    static_assert(H % 128 == 0 && H <= 1024, "hidden width must be a multiple of 128");
    constexpr int RS = H / 2;         // words per row of an A plane
    constexpr int NS = H / 128;       // 32-deep MFMA steps per K-quarter
    constexpr int NA = H / 128;       // float4 loads per thread that cover the 16 x H A tile
    constexpr int CT = H / 128;       // layer-0 column tiles per wave (variant A)
    constexpr int NT = H / 32;        // column tiles = head partials per row tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned* ahi = reinterpret_cast<unsigned*>(lds);   // [16][RS]
    unsigned* alo = ahi + 16 * RS;
    float* red = reinterpret_cast<float*>(alo + 16 * RS);   // [4 K-quarters][2 column tiles][64 lanes] float4
    float* ea = red + 4 * 2 * 64 * 4; // [16][32] actions of this step (variant A)
    float* hs = ea + 512;             // [16][36] GELU tile feeding the head partial (variant C)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int nt = wave & 1, kp = wave >> 1;
    int tm = blockIdx.x / NT, tn = blockIdx.x - tm * NT;
    if (P.xg) xcd_tile((int)blockIdx.x, P.M >> 4, NT, P.xg, tm, tn);
    const int row0 = tm * 16, n0 = tn * 32 + nt * 16;
    constexpr int variant = V;
    tl_enter(P.tl);
    // ---- every load of the launch that does not depend on another workgroup's data of THIS launch goes out first
    u32x4 bh[NS], bl[NS];
    {
        const unsigned* wb = reinterpret_cast<const unsigned*>(P.Wf) + (((size_t)(2 * kp * NS) * H + n0 + c) * 4 + q) * 4;
#pragma unroll
        for (int s = 0; s < NS; ++s) { bh[s] = ldg4u(wb + (size_t)s * H * 32); bl[s] = ldg4u(wb + (size_t)s * H * 32 + H * 16); }
    }
    const float bias = ldg(P.bias + n0 + c);
    u32x4 w4h, w4l;
    if (variant == 2 && wave == 0) {
        const unsigned* w4 = reinterpret_cast<const unsigned*>(P.W4f) + (((size_t)(2 * tn) * P.ap + c) * 4 + q) * 4;
        w4h = ldg4u(w4); w4l = ldg4u(w4 + P.ap * 16);
    }
    if (variant == 0) {
        // layer 0: C0 (loop invariant) + [a_s | t_s | 0] (16 x 16) times the 16 rows of W0 that start at the action block
        f32x4 cacc[CT];
        u32x4 wf[CT];   // hi 2 words | lo 2 words of k = 4q .. 4q + 3
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int ct = wave + 8 * t;
            cacc[t] = ldg4(P.A + (((size_t)(row0 >> 4) * (H / 16) + ct) * 64 + 16 * q + c) * 4);   // C0 in GF_C_FRAGT layout: row c, columns 16 ct + 4q .. + 3, 1 KB contiguous per tile
            wf[t] = ldg4u(reinterpret_cast<const unsigned*>(P.W0f) + ((size_t)q * H + 16 * ct + c) * 4);
        }
        {   // a_s = a_{s-1} + (sum of head partials + head bias) / flow_steps: two threads per element, fixed order
            const int e = tid >> 1, half = tid & 1;
            const int r = e >> 4, j = e & 15;
            float a = 0.f;
            if (j < P.ad) {
                a = ldg(P.ea_in + (size_t)(row0 + r) * P.ea_ld + j);
                if (P.evp_in) {
                    float pv[NT / 2];
                    const float* pp = P.evp_in + ((size_t)(half * (NT / 2)) * P.M + row0 + r) * P.ap + j;
#pragma unroll
                    for (int tp = 0; tp < NT / 2; ++tp) pv[tp] = ldg(pp + (size_t)tp * P.M * P.ap);
                    float sum = 0.f;
#pragma unroll
                    for (int tp = 0; tp < NT / 2; ++tp) sum += pv[tp];
                    const float other = __shfl_xor(sum, 1);
                    const float tot = half ? other + sum : sum + other;   // (partials 0..NT/2-1) + (NT/2..NT-1) on both lanes
                    a += (tot + ldg(P.eb + j)) * P.inv_steps;
                }
                if (tn == 0 && half == 0 && P.ea_out) stg(P.ea_out + (size_t)(row0 + r) * P.ap + j, a);
            } else if (j == P.ad) {
                a = P.t_s;
            }
            if (half == 0) ea[r * 32 + j] = a;
        }
        __syncthreads();
        {
            // layer 0 TRANSPOSED: D[n][row] = sum_k W0[k][n] A'[row][k] - the weight fragment is the A operand, the action fragment the B
            // operand (the same registers, swapped) - so a lane ends up with 4 CONSECUTIVE columns n = 16 ct + 4q + i of row c: its GELU
            // values are 2 hi + 2 lo words, one 8-byte LDS store per plane, no lane exchange
            const f32x4 af = *reinterpret_cast<const f32x4*>(&ea[c * 32 + 4 * q]);   // A'[row c][k = 4q + s]
            u32x2 afh, afl;
            bsplit4(af, afh, afl);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const u32x2 wh = u32x2{wf[t][0], wf[t][1]}, wl = u32x2{wf[t][2], wf[t][3]};
                cacc[t] = mfma_bf16_k16(wl, afh, cacc[t]);
                cacc[t] = mfma_bf16_k16(wh, afl, cacc[t]);
                cacc[t] = mfma_bf16_k16(wh, afh, cacc[t]);
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int ct = wave + 8 * t;
                f32x4 g;
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = gelu_f(cacc[t][i]);
                u32x2 hi, lo;
                bsplit4(g, hi, lo);
                const int w = c * RS + 4 * ((2 * ct + (q >> 1)) ^ c) + 2 * (q & 1);
                *reinterpret_cast<u32x2*>(ahi + w) = hi;
                *reinterpret_cast<u32x2*>(alo + w) = lo;
            }
        }
    } else {
        // 16 rows x RS words of each plane are contiguous: 16-byte pieces, NA / 2 per thread and plane
        u32x4 av[NA];
        const unsigned* Ah = reinterpret_cast<const unsigned*>(P.A) + (size_t)row0 * RS;
        const unsigned* Al = Ah + (size_t)P.M * RS;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pc = (tid + i * FQL_CHAIN_THREADS) & (4 * RS - 1);   // piece inside the plane: row pc / (RS / 4), slot pc % (RS / 4)
            av[i] = ldg4u((i < NA / 2 ? Ah : Al) + (size_t)pc * 4);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pc = (tid + i * FQL_CHAIN_THREADS) & (4 * RS - 1);
            const int r = pc / (RS / 4), sl = pc - r * (RS / 4);
            *reinterpret_cast<u32x4*>((i < NA / 2 ? ahi : alo) + r * RS + 4 * (sl ^ r)) = av[i];
        }
    }
    __syncthreads();
    // ---- this wave's K-quarter: NS steps x 3 MFMAs on three accumulator chains
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int w = c * RS + 4 * ((4 * (kp * NS + s) + q) ^ c);
        const u32x4 fah = *reinterpret_cast<const u32x4*>(ahi + w);
        const u32x4 fal = *reinterpret_cast<const u32x4*>(alo + w);
        acc = mfma_bf16(fah, bh[s], acc);
        acc2 = mfma_bf16(fah, bl[s], acc2);
        acc3 = mfma_bf16(fal, bh[s], acc3);
    }
    acc += acc2 + acc3;
    *reinterpret_cast<f32x4*>(&red[((kp * 2 + nt) * 64 + lane) * 4]) = acc;
    __syncthreads();
    // ---- epilogue shared by the 8 waves: wave (nt, kp) finishes row 4q + kp, column n0 + c of its column tile
    float v = bias;
#pragma unroll
    for (int p = 0; p < 4; ++p) v += red[((p * 2 + nt) * 64 + lane) * 4 + kp];
    v = gelu_f(v);
    const int row = row0 + 4 * q + kp;
    if (variant != 2) {
        const float o = lane_xor1(v);
        unsigned h, l;
        bsplit2((c & 1) ? o : v, (c & 1) ? v : o, h, l);
        unsigned* Cw = reinterpret_cast<unsigned*>(P.C) + ((c & 1) ? (size_t)P.M * RS : 0) + (size_t)row * RS + ((n0 + c) >> 1);
        *(FQL_GAS unsigned*)Cw = (c & 1) ? l : h;
    } else {
        hs[(4 * q + kp) * 36 + 16 * nt + c] = v;
        __syncthreads();
        if (wave == 0) {   // 16 x 32 GELU tile times this workgroup's 32 rows of the head kernel: one 32-deep step
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&hs[c * 36 + 8 * q]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&hs[c * 36 + 8 * q + 4]);
            u32x2 h0, l0, h1, l1;
            bsplit4(a0, h0, l0);
            bsplit4(a1, h1, l1);
            const u32x4 fah = u32x4{h0[0], h0[1], h1[0], h1[1]}, fal = u32x4{l0[0], l0[1], l1[0], l1[1]};
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
            pa = mfma_bf16(fal, w4h, pa);
            pa = mfma_bf16(fah, w4l, pa);
            pa = mfma_bf16(fah, w4h, pa);
            if (c < P.ap) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stg(P.evp_out + ((size_t)tn * P.M + row0 + 4 * q + i) * P.ap + c, pa[i]);
            }
        }
    }
    tl_exit(P.tl);
}
#define FQL_CHAIN_LDS_BYTES(H) ((size_t)(16 * ((H) + 4) + 4 * 2 * 64 * 4 + 512 + 16 * 36) * sizeof(float))
