// fql_xchain.h -- the Euler chain of the BC flow (agents/fql.py:155-171) as ONE persistent, XCD-resident launch (round 3).
//
// The chain is the critical path of an update: flow_steps x (layer 0, hidden layers 1 .. nh - 1, action head) strictly dependent
// [B x H] x [H x H] products.  As launches it was 30 dependent kernels per update, each 1.6 us of boundary plus a cold L2 in front of
// ~1 us of matrix work.  Here the batch is cut into 8 row blocks, one per XCD, and the 32 workgroups (one per CU) the dispatcher places on
// an XCD take that block through every layer of every step: a layer is split over the 32 members by output columns (16 each at H = 512), a
// member needs the whole [rows x H] activation panel of its block, and the panel is exchanged through THAT XCD's L2 only - plain stores
// (they stay in the L2), sc1 loads (they bypass the reader's L1), tile-major so that every wave load / store is one contiguous KB.  Rows never
// meet across XCDs (the chain is row-wise), so there is no cross-XCD traffic and no agent-scope fence inside the launch.  Each member streams
// its 16 columns of the next hidden kernel into registers right behind its arrival (eight 16-byte loads a lane from the fragment-major copies
// fql_wfrag_kernel keeps, in flight while it waits for the others) and keeps its rows of the action head and of layer 0's rank-16 update and its
// tile of C0 = obs W0[obs rows] + b0 in registers; the launch takes 16 KB of LDS.  (With the kernels resident in LDS instead - 3 x 32 KB a
// member - the chain alone is faster, 136 us, but a CU then has 45 KB of LDS left and the update takes 393 us against 352 for this form.)
//
// Synchronisation: member m of XCD g stores its phase count into word m of that XCD's flag line (a plain store: L2), a waiting workgroup polls
// the 32 words with one 32-lane sc1 load.  (An agent-scope atomic counter executes at the memory side, not in the L2: measured 1.4 us from the
// last arrival to the poll that sees it, against 0.33 us for the flags.)  Groups are formed from HW_REG_XCC_ID - the XCD a workgroup really
// runs on - and members by a per-XCD ticket, so the result never depends on which workgroup landed where; what the kernel needs is 32 resident
// workgroups per XCD (a 256-workgroup grid at one per CU; it leaves 128 registers a lane and ~45 KB of LDS per CU to the side lanes' kernels
// that run beside it).  A group that is short of members times out in its first wait (every spin is bounded), sets the sticky error word and
// drains; the optimizer launches then leave the parameters alone and the host reports it when the infos are read.
//
// Measured (experiments/xcd_phase.hip, the microbenchmark this was priced with): 2.7-2.9 us per dependent 512 x 512 layer inside the launch
// against 4.6 us per launch alone on the chip and ~9 us beside the side lanes.  The same structure for EVERY pass of the update (one XCD =
// 32 rows through all four networks) was built and measured too - correct, 1.5x slower than launches: with 32 rows per XCD every weight is
// fetched eight times per update and every op is a latency chain (git history of this round).
#pragma once

#define XCH_NMEM 32
#define XCH_NGRP 8
#define XCH_MAXRT 8            // 16-row tiles per XCD (batch <= 1024)

struct XChainArgs {
    int B, R, RT;              // batch, rows per XCD, 16-row tiles per XCD
    int H, nl;                 // hidden width of the BC flow; hidden kernels 1 .. nl
    int od, ad, ap, in_p, fs;
    const float* x_e0;         // [B, in_p] (obs | 0): C0's input
    const float* x_eu;         // [B, in_p] (obs | z | 0): initial actions
    const float* w0;           // first kernel [in_p][H]
    const float* b0;
    const float* wf[7];        // hidden kernels 1 .. nl as fragment-major copies [H/4][H][4] (fql_wfrag_kernel)
    const float* b[7];
    const float* w4;           // action head [H][ap]
    const float* b4;
    float* hc[2];              // activation panels [B, H], tile-major: [row tile][column tile][quad][row][4]
    float* vp;                 // head partials [32 members][B][16]
    float* tgt;                // [B, ap] out: clip(Euler result)  (agents/fql.py:170)
    unsigned* sync;            // 32-word slots: [0, 8) arrival flags per XCD, [8, 16) tickets, [16] error word (sticky); all but the last zeroed by the prep launch
    unsigned long long* stamps; // diagnostics build (-DFQL_XSTAMPS): [256][phases][2] s_memrealtime ticks: wait over, arrived
};

__device__ __forceinline__ f32x4 xc_ldx4(const float* base, unsigned off) {   // 16-byte load that bypasses this CU's L1 (sc1): data another CU of the XCD wrote
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4u), 0, 16));
}
__device__ __forceinline__ unsigned xc_ldx1(const unsigned* base, unsigned off) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4u), 0, 16));
}

struct XChCtx {
    int g, member, wave, lane, r, q;
    unsigned phase;            // phases this member has completed
    unsigned* flags;           // this XCD's flag line
    FQL_GAS unsigned* err;
    unsigned* dead;            // LDS word: a wait timed out
};

// every member of this XCD has completed c.phase phases; true = timed out
__device__ __forceinline__ bool xc_wait(XChCtx& c) {
    if (c.wave == 0) {
        unsigned spins = 0;
        for (;;) {
            const bool ok = c.lane >= XCH_NMEM || xc_ldx1(c.flags, (unsigned)c.lane) >= c.phase;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (c.lane == 0) { __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *c.dead = 1u; }
                break;
            }
        }
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int)*c.dead) != 0;
}
__device__ __forceinline__ void xc_arrive(XChCtx& c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this phase has reached the L2 ...
    __syncthreads();
    ++c.phase;
    if (threadIdx.x == 0) stg(reinterpret_cast<float*>(c.flags) + c.member, __builtin_bit_cast(float, c.phase));   // ... before this member counts as arrived
}
// offset (floats) of lane (r, q)'s 16 bytes of tile (row base rb, column tile ct) of a tile-major [rows, 16 ntn] tensor
__device__ __forceinline__ unsigned xc_toff(int lane, int rb, int ct, int ntn) { return (unsigned)((((rb >> 4) * ntn + ct) << 8) + (lane << 2)); }

__global__ __launch_bounds__(256, 3) void fql_xchain_kernel(const XChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xch_lds[];
    XChCtx c;
    c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.r = c.lane & 15; c.q = c.lane >> 4;
    c.g = (int)(__builtin_amdgcn_s_getreg(6164) & 7u);   // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
    c.phase = 0u;
    c.flags = a.sync + 32 * c.g;
    c.err = (FQL_GAS unsigned*)(a.sync + 32 * 16);
    f32x4* red = reinterpret_cast<f32x4*>(xch_lds);                 // [4 waves][2 tiles][64] float4
    float* alds = xch_lds + 8 * 64 * 4;                               // [R][16] current actions (+ t column)
    unsigned* misc = reinterpret_cast<unsigned*>(alds + XCH_MAXRT * 16 * 16);
    c.dead = misc + 1;
    if (threadIdx.x == 0) {
        misc[0] = __hip_atomic_fetch_add((FQL_GAS unsigned*)(a.sync + 32 * (8 + c.g)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[1] = 0u;
    }
    __syncthreads();
    c.member = __builtin_amdgcn_readfirstlane((int)misc[0]);   // (an LDS read: uniform, but only this tells the compiler)
    if (c.member >= XCH_NMEM) {   // more than 32 workgroups on this XCD: not a placement this kernel runs on
        if (threadIdx.x == 0) __hip_atomic_store(c.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int H = a.H, J = H >> 4, n0 = 16 * c.member, RT = a.RT, R = a.R;
    const bool active = n0 < H;
    const int rbase = c.g * R;
    // rows od .. od + 15 of the first kernel (the action block, t, zero padding) and this member's 16 rows of the action head, as first MFMA operands
    f32x4 w0f, w4f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = a.od + 4 * c.q + t, k4 = n0 + 4 * c.q + t;
        w0f[t] = (active && k < a.in_p) ? ldg(a.w0 + (size_t)k * H + n0 + c.r) : 0.f;
        w4f[t] = (active && k4 < H && c.r < a.ap) ? ldg(a.w4 + (size_t)k4 * a.ap + c.r) : 0.f;
    }
    // C0 = obs W0[obs rows] + b0 for this member's columns: loop invariant over the flow steps, kept in registers (wave w finishes tiles w, w + 4)
    f32x4 c0v[2];
    {
        const int JI = a.in_p >> 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            c0v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int t = c.wave + 4 * u;
            if (active && t < RT) {
                f32x4 acc = ldg4(a.b0 + n0 + 4 * c.q);
                for (int j = 0; j < JI; ++j) {
                    const f32x4 av = ldg4(a.x_e0 + (size_t)(rbase + 16 * t + c.r) * a.in_p + 16 * j + 4 * c.q);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int k = 16 * j + 4 * c.q + tt;
                        const float wv = k < a.od ? ldg(a.w0 + (size_t)k * H + n0 + c.r) : 0.f;   // (the action / t rows belong to the per-step update)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av[tt], acc, 0, 0, 0);
                    }
                }
                c0v[u] = acc;
            }
        }
    }
    // a_0 = the noise z
    for (int e = threadIdx.x; e < R * 16; e += 256) {
        const int row = e >> 4, col = e & 15;
        alds[e] = col < a.ad ? ldg(a.x_eu + (size_t)(rbase + row) * a.in_p + a.od + col) : 0.f;
    }
    __syncthreads();
    const int nq = (a.ad + 3) >> 2;   // live column quads of a head partial
    const float inv = 1.0f / (float)a.fs;
    // a_s = a_{s-1} + (sum of the 32 members' head partials of step s - 1 + head bias) / flow_steps, t column := s / flow_steps
    // (agents/fql.py:166-169); every member folds for itself and keeps the actions of its XCD's rows in LDS
    auto fold = [&](int s) {
        for (int t = 0; t < RT; ++t) {
            f32x4 pa{0.f, 0.f, 0.f, 0.f};
            if (c.q < nq) {
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) pa += xc_ldx4(a.vp, (unsigned)(((size_t)(c.wave + 4 * mm) * a.B + rbase + 16 * t + c.r) * 16 + 4 * c.q));
            }
            __syncthreads();
            red[c.wave * 64 + c.lane] = pa;
            __syncthreads();
            if (c.wave == 0) {
                f32x4 v = red[c.lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += red[w * 64 + c.lane];
                float* ar = alds + (16 * t + c.r) * 16 + 4 * c.q;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int col = 4 * c.q + tt;
                    if (col < a.ad) ar[tt] = ar[tt] + (v[tt] + ldg(a.b4 + col)) * inv;
                    else if (col == a.ad) ar[tt] = (float)s * inv;
                }
            }
        }
        __syncthreads();
    };
    // the fragment of hidden layer l (1-based) for this wave's K slices, from memory as flax stores it: lane (n, q) of slice j holds W[16 j + 4 q + t][n0 + n]
    f32x4 wreg[8];
    auto wfetch = [&](int l) {
        if (!active) return;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = c.wave + 4 * jj;
            wreg[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < J) wreg[jj] = ldg4(a.wf[l - 1] + ((size_t)(4 * j + c.q) * H + n0 + c.r) * 4);   // W[16 j + 4 q + t][n0 + n], t = 0..3: one 16-byte load
        }
    };
#ifdef FQL_XSTAMPS
    int sp = 0;
#define XCH_ST(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[((size_t)blockIdx.x * 64 + (sp & 63)) * 2 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XCH_ST(k) do {} while (0)
#endif
    for (int s = 0; s < a.fs; ++s) {
        if (s > 0) {
            if (xc_wait(c)) return;
            XCH_ST(0);
            fold(s);
        }
        // layer 0: GELU(C0 + [a_s | t_s] W0[act rows, t row]); K = 16, so one wave finishes a row tile by itself
        if (active) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = c.wave + 4 * u;
                if (t < RT) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(alds + (16 * t + c.r) * 16 + 4 * c.q);
                    f32x4 acc = c0v[u];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[tt], av[tt], acc, 0, 0, 0);
                    f32x4 g;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) g[tt] = gelu_f(acc[tt]);
                    stg4(a.hc[0] + xc_toff(c.lane, rbase + 16 * t, c.member, J), g);
                }
            }
        }
        xc_arrive(c);
        wfetch(1);
#ifdef FQL_XSTAMPS
        XCH_ST(1); ++sp;
#endif
        for (int l = 1; l <= a.nl; ++l) {
            if (xc_wait(c)) return;
            XCH_ST(0);
            if (active) {
                const float* A = a.hc[(l - 1) & 1];
                float* Co = a.hc[l & 1];
                const f32x4 bv = ldg4(a.b[l - 1] + n0 + 4 * c.q);
                for (int t0 = 0; t0 < RT; t0 += 2) {   // two row tiles per pass share each weight-fragment read; K is split over the four waves
                    const bool two = t0 + 1 < RT;
                    f32x4 av[2][8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = c.wave + 4 * jj;
                        av[0][jj] = av[1][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (j < J) {
                            av[0][jj] = xc_ldx4(A, xc_toff(c.lane, rbase + 16 * t0, j, J));
                            if (two) av[1][jj] = xc_ldx4(A, xc_toff(c.lane, rbase + 16 * t0 + 16, j, J));
                        }
                    }
                    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const f32x4 wv = wreg[jj];
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[tt], av[0][jj][tt], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[tt], av[1][jj][tt], acc[1], 0, 0, 0);
                        }
                    }
                    __syncthreads();
                    red[(c.wave * 2 + 0) * 64 + c.lane] = acc[0];
                    red[(c.wave * 2 + 1) * 64 + c.lane] = acc[1];
                    __syncthreads();
                    if (c.wave < 2 && t0 + c.wave < RT) {
                        f32x4 v = red[c.wave * 64 + c.lane];
#pragma unroll
                        for (int w = 1; w < 4; ++w) v += red[(w * 2 + c.wave) * 64 + c.lane];
                        v += bv;
                        f32x4 g;
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) g[tt] = gelu_f(v[tt]);
                        const int rb = rbase + 16 * (t0 + c.wave);
                        if (l < a.nl) stg4(Co + xc_toff(c.lane, rb, c.member, J), g);
                        else {   // last hidden layer: this member's 16 columns times its 16 rows of the action head -> a partial of the velocity
                            f32x4 pv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) pv = __builtin_amdgcn_mfma_f32_16x16x4f32(w4f[tt], g[tt], pv, 0, 0, 0);
                            if (4 * c.q < a.ad) stg4(a.vp + ((size_t)c.member * a.B + rb + c.r) * 16 + 4 * c.q, pv);
                        }
                    }
                }
            }
            xc_arrive(c);
            if (l < a.nl) wfetch(l + 1);
#ifdef FQL_XSTAMPS
            XCH_ST(1); ++sp;
#endif
        }
    }
    // the last step's velocity, then clip (agents/fql.py:170): member 0 of each XCD stores its block's rows
    if (xc_wait(c)) return;
    fold(a.fs);
    if (c.member == 0)
        for (int e = threadIdx.x; e < R * 4; e += 256) {
            const int row = e >> 2, qd = e & 3;
            f32x4 v;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) v[tt] = (4 * qd + tt < a.ad) ? clip1(alds[row * 16 + 4 * qd + tt]) : 0.f;
            if (4 * qd < a.ap) stg4(a.tgt + (size_t)(rbase + row) * a.ap + 4 * qd, v);
        }
}

#define FQL_XCHAIN_LDS_FLOATS (8 * 64 * 4 + XCH_MAXRT * 16 * 16 + 16)
