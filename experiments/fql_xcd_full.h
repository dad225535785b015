// fql_xcd.h -- the XCD-resident update kernel (round 3).
//
// One persistent launch runs every forward pass and every input-gradient pass of one FQL update (agents/fql.py:22-92, 155-171;
// utils/networks.py:34-61): the batch is cut into 8 row blocks, one per XCD, and the 32 workgroups (one per CU) that the dispatcher
// places on an XCD take that block through every layer.  A layer is split over the 32 members by output columns (16 each at H = 512), so
// a member needs the whole [rows x K] activation panel of its block: it is exchanged through THAT XCD's L2 only (plain stores, which
// stay in the L2; loads with sc1, which bypass the reader's L1), behind one arrival counter per XCD.  There is no kernel boundary
// between dependent layers (1.6 us + a cold L2 each, 82 of them per update in the launch-per-level program this replaces), no cross-XCD
// traffic inside the launch, and the Euler chain's hidden kernels stay in LDS for all flow_steps.  Rows never meet across XCDs inside the
// launch: every loss of the path is a mean over independent rows (agents/fql.py:37,59,66,73); what does reduce over the batch - weight
// gradients, LayerNorm scale / bias gradients, the info scalars - is left to the launches behind this one (per-XCD partials for the
// scalars).  experiments/xcd_phase.hip is the microbenchmark the design was priced with (2.7 us per dependent 512 x 512 layer).
//
// Groups are formed from HW_REG_XCC_ID - the XCD a workgroup really runs on - and members by a per-XCD ticket, so the result never
// depends on which workgroup landed where; what the kernel does need is 32 resident workgroups per XCD (a 256-workgroup grid at one per
// CU).  A group that is short of members times out in its first wait, sets the error word and drains (every spin is bounded); the
// optimizer launch behind it then leaves the parameters alone and the host reports the failure when the infos are read.
//
// Layout.  Inside the launch every [rows, H] tensor (activations, GELU', gradients) lives TILE-MAJOR: 16 x 16 tiles, a tile the 1 KB
// [column quad q][row r][4 columns] - the order the MFMA operand of the next layer wants, so a wave loads or stores a tile as one fully
// coalesced dwordx4 instruction (row-major buffers cost four times the L2 requests: the 16 lanes of a quarter wave sit in 16 different
// rows).  What the launches BEHIND this one read (weight gradients, LayerNorm scale / bias gradients: X, dZ, dY, g, row statistics) is
// also stored row-major, one extra store per epilogue.  16-wide tensors (heads, actions) are row-major only: a 16 x 16 tile of a
// 16-wide matrix is contiguous either way.
//
// The product MFMA is v_mfma_f32_16x16x4_f32 in its "transposed" use: weights are the first operand, activations the second, so a
// lane (r, q) of the accumulator holds out[row r][columns 4 q .. 4 q + 3] - exactly the 16 bytes the NEXT layer's lane loads as its
// operand, and one dwordx4 store per lane in the epilogue.
#pragma once

#define XCD_NMEM 32
#define XCD_NGRP 8
#define XCD_MAXROWS 128        // rows per XCD (batch <= 1024)

enum : int { XK_FWD = 0, XK_HEAD, XK_DGRAD, XK_SEED_DGRAD, XK_LNBWD, XK_CHAIN_L0, XK_CHAIN_MID, XK_CHAIN_LAST, XK_METRICS };
enum : int {
    XF_GELU = 1 << 0,        // GELU-tanh epilogue (utils/networks.py:46,56)
    XF_SAVEZ = 1 << 1,       // also store GELU'(z) (backward multiplies by it: no transcendental there)
    XF_BIAS = 1 << 2,
    XF_LN = 1 << 3,          // LayerNorm the A panel first (utils/networks.py:58; flax fast variance, eps 1e-6)
    XF_LN_STORE = 1 << 4,    // ... and store LN(A) (this member's columns) and the row statistics for the backward pass
    XF_ZMUL = 1 << 5,        // dgrad epilogue: C = acc * GELU'(z) of the layer below
    XF_OS_SCATTER = 1 << 6,  // one-step head on [next_obs ; obs ; obs] rows: clip(out) into the critic inputs (agents/fql.py:26,69)
    XF_SEED_ACTOR = 1 << 7,  // XK_SEED_DGRAD: A = d(actor loss)/d(one-step actions) (agents/fql.py:66-79), after folding the last Euler step
    XF_SYN = 1 << 8,         // XK_LNBWD: dY = dq (x) w (scalar head, rank 1)
    XF_SEED_CRITIC = 1 << 9, // XK_LNBWD + XF_SYN: dq = (q - y) / B computed here (agents/fql.py:28-37) and stored for the head's weight gradient
    XF_A_TILE = 1 << 10,     // A is tile-major [rows / 16][K / 16][4][16][4] (else row-major with lda)
};

struct XOp {
    int kind, flags;
    const float* A;      // [rows, lda] activations (XK_LNBWD: dY)
    const float* W;      // kernel [in][out] row-major as flax stores it (forward: in = K, dgrad: in = N)
    const float* bias;
    float* C;            // [rows, ldc] row-major copy of the output (null: none)
    float* Ct;           // tile-major copy of the output (null: none)
    float* Zout;         // XF_SAVEZ: GELU'(z), tile-major
    const float* Zmul;   // XF_ZMUL / XK_LNBWD: stored GELU'(z), tile-major
    const float *ln_g, *ln_b;
    float* ln_xout;      // XF_LN_STORE
    float* ln_stats;     // XF_LN_STORE: [rows, 2] mean, rstd (XK_LNBWD: read)
    int lda, ldw, ldc, K, N;
    int nblk, blk_stride;   // stacked row blocks (one-step actor: 3 blocks of B rows)
    int member0;            // column tile t (or LNBWD row) is taken by member (t + member0) % 32
    int step;               // chain ops: Euler step
    // kind-specific operands
    const float *p0, *p1, *p2, *p3, *p4, *p5;
    float *o0, *o1;
    int i0, i1;
    float f0, f1;
};

// Teams.  A workgroup is 16 waves = four teams of four, each running its own program; on every SIMD one wave of each team is resident,
// so while one team waits for its loads or its XCD barrier the others have the matrix pipe:
//   team 0: C0, the Euler chain of the even row tiles, then the one-step actor's backward (the critical path of the update)
//   team 1: the Euler chain of the odd row tiles (a second, independent pipeline: rows do not interact)
//   teams 2, 3: everything else, one shared phase list; team 2 runs ops [first, first + count_a) of a phase, team 3 the rest.
// A phase of a team starts when every member of its XCD has finished the previous phase of that team (teams 2 and 3: of both) and wait[k]
// phases of team k.  chain > 0: the phase opens the Euler chain - `chain` phases of one chain op each, run on a dedicated path.
struct XPhase { int first, count, count_a, chain; int wait[4]; };

struct XcdArgs {
    const XOp* ops;
    const XPhase* phases;    // team 0's phases [0, nphase0), then the phases of teams 2 / 3 [nphase0, nphase0 + nphase1)
    int nphase0, nphase1;
    int nct;                 // chain pipelines: 2 = team 1 takes the odd row tiles, 1 = team 0 takes all
    int chain_p0;            // team 0's phase index of the first chain phase
    int B, R, RT;            // batch, rows per XCD, 16-row tiles per XCD
    unsigned* sync;          // slots of 32 words: slot 8 t + g = team t's arrival flags on XCD g (one word per member), slots 32 + g tickets, slot 40 the
                             // error flag (sticky); all but the error flag zeroed by the prep launch
    float* xpart;            // [8][16] per-XCD partial sums of the info scalars
    // Euler chain (agents/fql.py:155-171)
    const float* chain_w[7]; // hidden kernels 1 .. nh - 1 of the BC flow, [H][H]
    const float* chain_b[7]; // their biases
    float* hc[2];            // the chain's activation buffers (tile-major, ping-pong)
    const float* c0;         // obs W0[obs rows] + b0 (tile-major), loop invariant
    int chain_nl;            // how many of them (LDS resident: chain_nl * H * 64 bytes)
    int H;                   // hidden width of the BC flow
    const float* w0;         // its first kernel [in_p][H]: rows od .. od + 15 are the rank-16 update of layer 0
    const float* w4;         // its head kernel [H][ap]
    const float* b4;         // head bias
    const float* x_eu;       // [B, in_p] (obs | z | 0): initial actions
    float* vp;               // head partials [32][B][16]
    float* tgt;              // [B, ap] clip(Euler result) (written by member 0 for the metrics / debugging)
    int od, ad, ap, in_p, fs;
    int lds_floats;
    unsigned long long* stamps2;  // diagnostics build: [256][16] stamps inside the ops of phase `stamp_phase` of team `stamp_team` (+ [256] placement words)
    int stamp_phase, stamp_team, stamp_stride;
    int skip_team;                // diagnostics: ops of this team are not executed (results invalid); -1 = none
    unsigned long long* stamps;   // diagnostics build (-DFQL_XSTAMPS): [256 workgroups][2 teams][stamp_stride phases][4] s_memrealtime ticks (10 ns): wait over, ops done, drained, arrived
};

__device__ __forceinline__ f32x4 ldx4(const float* base, unsigned off) {   // 16-byte load that bypasses this CU's L1 (sc1): data another CU of the XCD wrote
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4u), 0, 16));
}
__device__ __forceinline__ float ldx1(const float* base, unsigned off) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4u), 0, 16));
}

struct XCtx {
    int g, member, wave, lane, r, q;   // wave: 0..3 inside the team
    int team, tid;                     // team 0..3; tid: 0..255 inside the team
    int t_first, t_step;               // chain ops: the row tiles this team takes
    int R, RT;
    unsigned bt;                       // this wave's count of team barriers (x 4)
    __attribute__((address_space(3))) unsigned* bar;   // the team's barrier word in LDS
    f32x4* red;     // [4 waves][2 tiles][64] cross-wave reduction
    float* stat;    // [4 waves][2 tiles][16 rows][2]
    float* alds;    // [R][16] current Euler actions (+ t column)
    const f32x4* wlds;
    unsigned long long* st2;
    unsigned long long* stl;   // 16 stamps in LDS
    unsigned long long tw[6];  // diagnostics: ticks spent in poll / barrier after poll / arrive-drain / arrive-barrier / count
};
#ifdef FQL_XSTAMPS
// stamps go to LDS (flushed at the end of the phase): a global store per stamp would put its round trip into the next stamp's wait
#define XST(c, k) do { if ((c).st2 && (c).tid == 0) { if ((k) == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); (c).stl[k] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define XST(c, k) do {} while (0)
#endif

// Barrier of the four waves of a team.  A workgroup holds two teams that run different programs, so s_barrier (all eight waves) is
// of no use: the waves count up a word in LDS and spin on it.  LDS operations of a wave complete in order, so the lgkmcnt(0) wait in
// front makes everything it wrote to LDS visible before it counts as arrived.
__device__ __forceinline__ void xsync(XCtx& c) {
    c.bt += 4u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c.lane == 0) __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < c.bt) __builtin_amdgcn_s_sleep(1);   // (a bare spin takes LDS and issue slots from the other team)
    asm volatile("" ::: "memory");
}

// row base (global row of the [rows, ld] buffers) of tile ti of an op
__device__ __forceinline__ int xrow(const XCtx& c, const XOp& o, int ti) {
    const int blk = ti / c.RT, t = ti - blk * c.RT;
    return blk * o.blk_stride + c.g * c.R + 16 * t;
}

// ---- the dense core: out[rows of this XCD][16 columns of this member] ------------------------------------------------------------
// K is split over the four waves (wave w takes the 16-deep slices j = w, w + 4, ...: NJ of them, compile time); NT row tiles per pass
// (compile time: a conditional MFMA makes the compiler shuttle accumulators between register files around every product) share a
// weight fragment; partial accumulators meet in LDS and wave u finishes row tile u.  Slices beyond K load zeros.
template <int NT, int NJ, class RowOf, class ALoad, class Pro, class WLoad, class Epi>
__device__ __forceinline__ void xdense_nt(XCtx& c, int K, int N, int ntl, int mrel, RowOf rowof, ALoad aload, Pro pro, WLoad wload, Epi epi) {
    const int J = K >> 4, ntn = N >> 4;
    for (int ct = mrel; ct < ntn; ct += XCD_NMEM) {
        const int n0 = ct << 4;
        XST(c, 1);
        f32x4 wv[NJ];   // this member's weight fragment: loaded once, used for every row tile of the op
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = c.wave + 4 * jj;
            wv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < J) wv[jj] = wload(n0, j);
        }
        for (int t0 = 0; t0 < ntl; t0 += NT) {
            int rb[NT];   // row bases of the tiles of this pass: wave-uniform
#pragma unroll
            for (int u = 0; u < NT; ++u) rb[u] = rowof(t0 + u);
            f32x4 av[NT][NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = c.wave + 4 * jj;
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    av[u][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (j < J) av[u][jj] = aload(rb[u], j);
                }
            }
            XST(c, 2);
            pro(av, rb, J, ct);
            XST(c, 3);
            f32x4 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[jj][t], av[u][jj][t], acc[u], 0, 0, 0);
            XST(c, 4);
            xsync(c);
#pragma unroll
            for (int u = 0; u < NT; ++u) c.red[(c.wave * 2 + u) * 64 + c.lane] = acc[u];
            xsync(c);
            XST(c, 5);
            if (c.wave < NT) {
                f32x4 s = c.red[(0 * 2 + c.wave) * 64 + c.lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) s += c.red[(w * 2 + c.wave) * 64 + c.lane];
                int rbw = rb[0];
#pragma unroll
                for (int u = 1; u < NT; ++u) if (c.wave == u) rbw = rb[u];
                epi(rbw, n0, s);
            }
            XST(c, 6);
        }
    }
}
template <int NJ, class RowOf, class ALoad, class Pro, class WLoad, class Epi>
__device__ __forceinline__ void xdense(XCtx& c, int K, int N, int ntl, int mrel, RowOf rowof, ALoad aload, Pro pro, WLoad wload, Epi epi) {
    // one row tile per pass: 16 waves per CU leave 128 registers a lane, the panel of one tile and the weight fragment take 64 of them
    xdense_nt<1, NJ>(c, K, N, ntl, mrel, rowof, aload, pro, wload, epi);
}

struct XNoPro {
    template <int NT, int NJ>
    __device__ __forceinline__ void operator()(f32x4 (&)[NT][NJ], const int (&)[NT], int, int) const {}
};

// LayerNorm of the A panel in registers (utils/networks.py:58): every member needs every row's statistics and has the whole row
// (a K-quarter per wave), so each computes them itself - no partial-sum buffers, no extra phase.
struct XLnPro {
    XCtx& c;
    const XOp& o;
    template <int NT, int NJ>
    __device__ __forceinline__ void operator()(f32x4 (&av)[NT][NJ], const int (&rb)[NT], int J, int ct) const {
        float s1[NT], s2[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            s1[u] = s2[u] = 0.f;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int t = 0; t < 4; ++t) { const float v = av[u][jj][t]; s1[u] += v; s2[u] += v * v; }   // (slices beyond J and tiles beyond nt are zeros)
            s1[u] += __shfl_xor(s1[u], 16); s1[u] += __shfl_xor(s1[u], 32);
            s2[u] += __shfl_xor(s2[u], 16); s2[u] += __shfl_xor(s2[u], 32);
        }
        xsync(c);   // (the statistics slots of the previous pass have been read)
        if (c.q == 0) {
#pragma unroll
            for (int u = 0; u < NT; ++u) { c.stat[((c.wave * 2 + u) * 16 + c.r) * 2] = s1[u]; c.stat[((c.wave * 2 + u) * 16 + c.r) * 2 + 1] = s2[u]; }
        }
        xsync(c);
        const float inv = 1.0f / (float)o.K;
        const bool store_all = (o.N >> 4) == 1;   // a single-tile op (the scalar / action heads): its one member stores the whole normalised panel
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += c.stat[((w * 2 + u) * 16 + c.r) * 2]; a2 += c.stat[((w * 2 + u) * 16 + c.r) * 2 + 1]; }
            const float mean = a1 * inv;
            const float var = fmaxf(a2 * inv - mean * mean, 0.0f);
            const float rstd = 1.0f / sqrtf(var + 1e-6f);
            const int row = rb[u] + c.r;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int j = c.wave + 4 * jj;
                if (j < J) {
                    const f32x4 gm = ldg4(o.ln_g + 16 * j + 4 * c.q), bt = ldg4(o.ln_b + 16 * j + 4 * c.q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[u][jj][t] = (av[u][jj][t] - mean) * rstd * gm[t] + bt[t];
                    if ((o.flags & XF_LN_STORE) && (j == ct || store_all)) stg4(o.ln_xout + (size_t)row * o.lda + 16 * j + 4 * c.q, av[u][jj]);
                }
            }
            if ((o.flags & XF_LN_STORE) && ct == 0 && c.wave == 0 && c.q == 0) {
                stg(o.ln_stats + 2 * (size_t)row, mean);
                stg(o.ln_stats + 2 * (size_t)row + 1, rstd);
            }
        }
    }
};

// ---- Euler chain helpers -----------------------------------------------------------------------------------------------------------
// a_s = a_{s-1} + (sum of the 32 members' head partials of step s - 1 + head bias) / flow_steps, t column := s / flow_steps
// (agents/fql.py:166-169); every member folds for itself (the partials are [32][rows][16] floats, of which act_dim columns are live) and
// keeps the actions of its XCD's rows in LDS.  s = 0: the noise z.
__device__ __forceinline__ void xchain_fold(XCtx& c, const XcdArgs& a, int s, int t_first, int t_step) {
    const int nq = (a.ad + 3) >> 2;   // live column quads of a partial row
    if (s == 0) {
        xsync(c);
        for (int t = t_first; t < c.RT; t += t_step)
            for (int e = c.tid; e < 256; e += 256) {
                const int row = 16 * t + (e >> 4), col = e & 15;
                c.alds[row * 16 + col] = col < a.ad ? ldg(a.x_eu + (size_t)(c.g * c.R + row) * a.in_p + a.od + col) : 0.f;
            }
        xsync(c);
        return;
    }
    const float inv = 1.0f / (float)a.fs;
    XST(c, 9);
    for (int t = t_first; t < c.RT; t += t_step) {
        f32x4 pa{0.f, 0.f, 0.f, 0.f};
        if (c.q < nq) {
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) {
                const int m = c.wave + 4 * mm;
                pa += ldx4(a.vp, (unsigned)(((size_t)m * a.B + c.g * c.R + 16 * t + c.r) * 16 + 4 * c.q));
            }
        }
        XST(c, 10);
        xsync(c);
        c.red[c.wave * 64 + c.lane] = pa;
        xsync(c);
        XST(c, 11);
        if (c.wave == 0) {
            f32x4 v = c.red[c.lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += c.red[w * 64 + c.lane];
            float* ar = c.alds + (16 * t + c.r) * 16 + 4 * c.q;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int col = 4 * c.q + tt;
                if (col < a.ad) ar[tt] = ar[tt] + (v[tt] + ldg(a.b4 + col)) * inv;
                else if (col == a.ad) ar[tt] = (float)s * inv;
            }
        }
    }
    xsync(c);
    XST(c, 12);
}

// ---- one op of a phase ---------------------------------------------------------------------------------------------------------------
// offset (floats) of lane (r, q)'s 16 bytes of tile (row base rb, column tile ct) of a tile-major [rows, 16 ntn] tensor
__device__ __forceinline__ unsigned xtoff(const XCtx& c, int rb, int ct, int ntn) { return (unsigned)((((rb >> 4) * ntn + ct) << 8) + (c.lane << 2)); }

__device__ __forceinline__ void xrun(XCtx& c, const XcdArgs& a, const XOp* op, const float (&w4f)[4]) {
    // the op table is constant for the launch: read through the constant address space, so every field is a scalar load into SGPRs
    // (as generic global loads they come back in VGPRs, and every buffer load whose descriptor is built from one gets a waterfall loop)
    XOp o;
    {
        const __attribute__((address_space(4))) unsigned* src = (const __attribute__((address_space(4))) unsigned*)op;
        unsigned* dst = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(XOp) / 4); ++i) dst[i] = src[i];
    }
    const int mrel = (c.member - o.member0) & (XCD_NMEM - 1);
    const int ntl = o.nblk * c.RT;
    XST(c, 0);
    auto rowof = [&](int ti) { return xrow(c, o, ti); };
    const int JA = o.K >> 4, NTN = o.N >> 4;
    auto aload = [&](int rb, int j) {
        if (o.flags & XF_A_TILE) return ldx4(o.A, xtoff(c, rb, j, JA));
        return ldx4(o.A, (unsigned)((size_t)(rb + c.r) * o.lda + 16 * j + 4 * c.q));
    };
    auto wfwd = [&](int n0, int j) {   // forward: lane (r, q) needs W[16 j + 4 q + t][n0 + r], t = 0..3
        f32x4 v;
        const float* p = o.W + (size_t)(16 * j + 4 * c.q) * o.ldw + n0 + c.r;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = ldg(p + (size_t)t * o.ldw);
        return v;
    };
    // both copies of an output tile: tile-major for the next op of this launch, row-major for the launches behind it
    auto store_out = [&](int rb, int n0, const f32x4& v) {
        if (o.Ct) stg4(o.Ct + xtoff(c, rb, n0 >> 4, NTN), v);
        if (o.C) stg4(o.C + (size_t)(rb + c.r) * o.ldc + n0 + 4 * c.q, v);
    };
    switch (o.kind) {
        case XK_FWD:
        case XK_HEAD: {
            auto epi = [&](int rb, int n0, f32x4 v) {
                const int row = rb + c.r;
                if (o.flags & XF_BIAS) v += ldg4(o.bias + n0 + 4 * c.q);
                if (o.flags & XF_GELU) {
                    f32x4 g, dg;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { float gg, dd; gelu_both(v[t], gg, dd); g[t] = gg; dg[t] = dd; }
                    store_out(rb, n0, g);
                    if (o.flags & XF_SAVEZ) stg4(o.Zout + xtoff(c, rb, n0 >> 4, NTN), dg);
                } else {
                    store_out(rb, n0, v);
                }
                if (o.flags & XF_OS_SCATTER) {   // o0 = X_ct, o1 = X_c2; i0 = their ld, i1 = obs_dim
                    const int blk = rb >= 2 * o.blk_stride ? 2 : (rb >= o.blk_stride ? 1 : 0);
                    float* dst = blk == 0 ? o.o0 : (blk == 1 ? o.o1 : nullptr);
                    if (dst) {
                        const int brow = row - blk * o.blk_stride;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (n0 + 4 * c.q + t < a.ad) stg(dst + (size_t)brow * o.i0 + o.i1 + n0 + 4 * c.q + t, clip1(v[t]));
                    }
                }
            };
            if (o.flags & XF_LN) xdense<8>(c, o.K, o.N, ntl, mrel, rowof, aload, XLnPro{c, o}, wfwd, epi);
            else if (o.K <= 64) xdense<1>(c, o.K, o.N, ntl, mrel, rowof, aload, XNoPro{}, wfwd, epi);
            else xdense<8>(c, o.K, o.N, ntl, mrel, rowof, aload, XNoPro{}, wfwd, epi);
            break;
        }
        case XK_DGRAD: {   // C = A W^T: out column n <-> input neuron, lane (r, q) needs W[n0 + r][16 j + 4 q ..]: one dwordx4
            auto wbwd = [&](int n0, int j) { return ldg4(o.W + (size_t)(n0 + c.r) * o.ldw + 16 * j + 4 * c.q); };
            auto epi = [&](int rb, int n0, f32x4 v) {
                if (o.flags & XF_ZMUL) v *= ldx4(o.Zmul, xtoff(c, rb, n0 >> 4, NTN));
                store_out(rb, n0, v);
            };
            xdense<8>(c, o.K, o.N, ntl, mrel, rowof, aload, XNoPro{}, wbwd, epi);
            break;
        }
        case XK_SEED_DGRAD: {
            // head dgrad of an actor with the loss gradient built in place of the A load (K = 16):
            //   BC flow (agents/fql.py:58-59):  d = 2 (pred - vel) / (B ad)                      p0 = pred, p1 = vel, f0 = 2 / (B ad)
            //   one-step actor (:66-79):        d = alpha 2 (mu - tgt) / (B ad) + [|mu| < 1] (dQa + dQb)
            //                                   p0 = mu rows (raw one-step output of the (obs, z) block), p2 / p3 = critic input gradients [B, i0],
            //                                   i1 = obs_dim, f0 = alpha 2 / (B ad); the Euler target is folded here first
            if (o.flags & XF_SEED_ACTOR) xchain_fold(c, a, a.fs, 0, 1);
            auto seed = [&](int rb, int) {
                const int row = rb + c.r;
                f32x4 d{0.f, 0.f, 0.f, 0.f};
                if (4 * c.q < a.ad) {
                    const f32x4 pv = ldx4(o.p0, (unsigned)((size_t)row * a.ap + 4 * c.q));
                    if (o.flags & XF_SEED_ACTOR) {
                        const float* ar = c.alds + (row - c.g * c.R) * 16 + 4 * c.q;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int col = 4 * c.q + t;
                            if (col < a.ad) {
                                const float tg = clip1(ar[t]);
                                float g = o.f0 * (pv[t] - tg);
                                if (pv[t] > -1.0f && pv[t] < 1.0f)
                                    g += ldx1(o.p2, (unsigned)((size_t)row * o.i0 + o.i1 + col)) + ldx1(o.p3, (unsigned)((size_t)row * o.i0 + o.i1 + col));
                                d[t] = g;
                            }
                        }
                    } else {
                        const f32x4 vv = ldg4(o.p1 + (size_t)row * a.ap + 4 * c.q);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (4 * c.q + t < a.ad) d[t] = o.f0 * (pv[t] - vv[t]);
                    }
                    if (mrel == 0) {
                        stg4(o.o0 + (size_t)row * a.ap + 4 * c.q, d);   // dz of the head: its weight gradient reads it
                        if ((o.flags & XF_SEED_ACTOR) && o.o1) {
                            f32x4 tg;
#pragma unroll
                            for (int t = 0; t < 4; ++t) tg[t] = (4 * c.q + t < a.ad) ? clip1(c.alds[(row - c.g * c.R) * 16 + 4 * c.q + t]) : 0.f;
                            stg4(o.o1 + (size_t)row * a.ap + 4 * c.q, tg);
                        }
                    }
                }
                return d;
            };
            auto wbwd = [&](int n0, int j) { return ldg4(o.W + (size_t)(n0 + c.r) * o.ldw + 16 * j + 4 * c.q); };
            auto epi = [&](int rb, int n0, f32x4 v) {
                if (o.flags & XF_ZMUL) v *= ldx4(o.Zmul, xtoff(c, rb, n0 >> 4, NTN));
                store_out(rb, n0, v);
            };
            xdense<1>(c, 16, o.N, ntl, mrel, rowof, seed, XNoPro{}, wbwd, epi);
            break;
        }
        case XK_LNBWD: {
            // LayerNorm backward x GELU' (utils/networks.py:56-58 reversed) of one 16-row tile per member (the columns over its four waves):
            //   dxhat = dY gamma ; dg = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) ; dZ = dg GELU'(z)
            // A = dY tile-major (or XF_SYN: p0 = dq [rows, i0], W = head kernel [H][ldw]); p1 = GELU output (tile-major); Zmul = GELU'(z); ln_stats;
            // ln_g; Ct / C = dZ
            if (mrel >= c.RT) break;
            const int rb = c.g * c.R + 16 * mrel, row = rb + c.r;
            const int H = o.N, J = H >> 4;
            float dqr = 0.f;
            if (o.flags & XF_SYN) {
                if (o.flags & XF_SEED_CRITIC) {   // p2 / p3 = q of the two target members, p4 = rewards, p5 = masks; o0 = dq out; f0 = discount, i1 = q_agg; f1 = 1 / B
                    const float ta = ldx1(o.p2, (unsigned)((size_t)row * 16)), tb = ldx1(o.p3, (unsigned)((size_t)row * 16));
                    const float nq = o.i1 ? fminf(ta, tb) : 0.5f * (ta + tb);
                    const float y = ldg(o.p4 + row) + o.f0 * ldg(o.p5 + row) * nq;
                    dqr = (ldx1(o.p0, (unsigned)((size_t)row * o.i0)) - y) * o.f1;
                    if (c.wave == 0 && c.q == 0) stg(o.o0 + (size_t)row * o.i0, dqr);
                } else {
                    dqr = ldg(o.p0 + (size_t)row * o.i0);
                }
            }
            const float mean = ldx1(o.ln_stats, 2u * (unsigned)row), rstd = ldx1(o.ln_stats, 2u * (unsigned)row + 1u);
            // two passes over the row (it does not fit the registers of a 16-wave workgroup): sums first, then the gradient
            auto dxhat = [&](int j, f32x4& xh) {   // returns dY gamma of lane (r, q)'s 4 columns of slice j, xh = their xhat
                const f32x4 gq = ldx4(o.p1, xtoff(c, rb, j, J));
                const f32x4 gm = ldg4(o.ln_g + 16 * j + 4 * c.q);
                f32x4 dd;
                if (o.flags & XF_SYN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[e] = dqr * ldg(o.W + (size_t)(16 * j + 4 * c.q + e) * o.ldw);
                } else dd = ldx4(o.A, xtoff(c, rb, j, J));
#pragma unroll
                for (int e = 0; e < 4; ++e) { xh[e] = (gq[e] - mean) * rstd; dd[e] *= gm[e]; }
                return dd;
            };
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = c.wave + 4 * jj;
                if (j < J) {
                    f32x4 xh;
                    const f32x4 d = dxhat(j, xh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1 += d[e]; s2 += d[e] * xh[e]; }
                }
            }
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            xsync(c);
            if (c.q == 0) { c.stat[(c.wave * 16 + c.r) * 2] = s1; c.stat[(c.wave * 16 + c.r) * 2 + 1] = s2; }
            xsync(c);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += c.stat[(w * 16 + c.r) * 2]; a2 += c.stat[(w * 16 + c.r) * 2 + 1]; }
            const float inv = 1.0f / (float)H;
            const float m1 = a1 * inv, m2 = a2 * inv;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = c.wave + 4 * jj;
                if (j < J) {
                    f32x4 xh;
                    const f32x4 d = dxhat(j, xh);
                    const f32x4 zz = ldx4(o.Zmul, xtoff(c, rb, j, J));
                    f32x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = rstd * (d[e] - m1 - xh[e] * m2) * zz[e];
                    if (o.Ct) stg4(o.Ct + xtoff(c, rb, j, J), ov);
                    if (o.C) stg4(o.C + (size_t)row * o.ldc + 16 * j + 4 * c.q, ov);
                }
            }
            break;
        }
        case XK_CHAIN_L0: {
            // layer 0 of Euler step s: GELU(C0 + [a_s | t_s] W0[act rows, t row]) with C0 = obs W0[obs rows] + b0 computed once
            // (A = C0, tile-major; Ct = the chain's activation buffer)
            xchain_fold(c, a, o.step, 0, 1);
            auto al = [&](int rb, int) { return *reinterpret_cast<const f32x4*>(c.alds + (rb - c.g * c.R + c.r) * 16 + 4 * c.q); };
            auto wl = [&](int n0, int) {
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = a.od + 4 * c.q + t;
                    v[t] = k < a.in_p ? ldg(a.w0 + (size_t)k * o.ldw + n0 + c.r) : 0.f;
                }
                return v;
            };
            auto epi = [&](int rb, int n0, f32x4 v) {
                v += ldx4(o.A, xtoff(c, rb, n0 >> 4, NTN));
                f32x4 g;
#pragma unroll
                for (int t = 0; t < 4; ++t) g[t] = gelu_f(v[t]);
                stg4(o.Ct + xtoff(c, rb, n0 >> 4, NTN), g);
            };
            xdense<1>(c, 16, o.N, ntl, mrel, rowof, al, XNoPro{}, wl, epi);
            break;
        }
        case XK_CHAIN_MID:
        case XK_CHAIN_LAST: {
            const int J = o.K >> 4;
            const f32x4* wl_base = c.wlds ? c.wlds + (size_t)o.i0 * J * 64 : nullptr;   // i0 = slot of this layer's kernel in LDS
            auto wl = [&](int n0, int j) {
                if (wl_base) return wl_base[j * 64 + c.lane];
                return wfwd(n0, j);
            };
            auto epi = [&](int rb, int n0, f32x4 v) {
                const int row = rb + c.r;
                v += ldg4(o.bias + n0 + 4 * c.q);
                f32x4 g;
#pragma unroll
                for (int t = 0; t < 4; ++t) g[t] = gelu_f(v[t]);
                if (o.kind == XK_CHAIN_MID) { stg4(o.Ct + xtoff(c, rb, n0 >> 4, NTN), g); return; }
                // last hidden layer: this member's 16 columns times its 16 rows of the action head -> a partial of the velocity
                f32x4 pv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) pv = __builtin_amdgcn_mfma_f32_16x16x4f32(w4f[t], g[t], pv, 0, 0, 0);
                if (4 * c.q < a.ad) stg4(a.vp + ((size_t)c.member * a.B + row) * 16 + 4 * c.q, pv);
            };
            xdense<8>(c, o.K, o.N, ntl, mrel, rowof, aload, XNoPro{}, wl, epi);
            break;
        }
        case XK_METRICS: {
            // per-XCD partial sums of the info scalars (agents/fql.py:39-44,85-92), folded by the finalize launch:
            //   0 sum (q - y)^2 (both members)  1 sum q  2 max q  3 min q  4 sum (pred - vel)^2  5 sum (mu - tgt)^2  6 sum q2  7 sum |q2|  8 sum (clip(mu3) - a)^2
            // p0 / p1 = critic q [B, 16], p2 / p3 = target q, p4 = rewards, p5 = masks; A = bc pred, W = vel; bias = one-step out [3B, ap];
            // Zmul = tgt; ln_g / ln_b = q2 of the two members; ln_xout unused; o0 = w_act as const; f0 = discount, i1 = q_agg
            if (mrel != 0 || c.wave != 0) break;
            float sl = 0.f, sq = 0.f, mx = -INFINITY, mn = INFINITY, sbc = 0.f, sdi = 0.f, sq2 = 0.f, sa2 = 0.f, sms = 0.f;
            for (int lr = c.lane; lr < c.R; lr += 64) {
                const int row = c.g * c.R + lr;
                const float ta = ldx1(o.p2, (unsigned)row * 16u), tb = ldx1(o.p3, (unsigned)row * 16u);
                const float nq = o.i1 ? fminf(ta, tb) : 0.5f * (ta + tb);
                const float y = ldg(o.p4 + row) + o.f0 * ldg(o.p5 + row) * nq;
                const float qa = ldx1(o.p0, (unsigned)row * 16u), qb = ldx1(o.p1, (unsigned)row * 16u);
                sl += (qa - y) * (qa - y) + (qb - y) * (qb - y);
                sq += qa + qb; mx = fmaxf(mx, fmaxf(qa, qb)); mn = fminf(mn, fminf(qa, qb));
                const float q2 = 0.5f * (ldx1(o.ln_g, (unsigned)row * 16u) + ldx1(o.ln_b, (unsigned)row * 16u));
                sq2 += q2; sa2 += fabsf(q2);
                for (int col = 0; col < a.ad; ++col) {
                    const float db = ldx1(o.A, (unsigned)(row * a.ap + col)) - ldg(o.W + (size_t)row * a.ap + col);
                    sbc += db * db;
                    const float dd = ldx1(o.bias, (unsigned)((a.B + row) * a.ap + col)) - ldx1(o.Zmul, (unsigned)(row * a.ap + col));
                    sdi += dd * dd;
                    const float dm = clip1(ldx1(o.bias, (unsigned)((2 * a.B + row) * a.ap + col))) - ldg(o.o0 + (size_t)row * a.ap + col);
                    sms += dm * dm;
                }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                sl += __shfl_xor(sl, off); sq += __shfl_xor(sq, off); sbc += __shfl_xor(sbc, off); sdi += __shfl_xor(sdi, off);
                sq2 += __shfl_xor(sq2, off); sa2 += __shfl_xor(sa2, off); sms += __shfl_xor(sms, off);
                mx = fmaxf(mx, __shfl_xor(mx, off)); mn = fminf(mn, __shfl_xor(mn, off));
            }
            if (c.lane == 0) {
                float* xp = a.xpart + 16 * c.g;
                stg(xp + 0, sl); stg(xp + 1, sq); stg(xp + 2, mx); stg(xp + 3, mn); stg(xp + 4, sbc); stg(xp + 5, sdi); stg(xp + 6, sq2); stg(xp + 7, sa2); stg(xp + 8, sms);
            }
            break;
        }
    }
}

// Arrival flags instead of a counter: member m stores its phase count into word m of its team's flag line on its XCD (a plain store: it
// stays in the XCD's L2, like the activations), and a waiting team polls the 32 words with ONE 32-lane sc1 load of that line.  An
// agent-scope atomic executes at the memory side, not in the L2: counter arrivals measured 1.4 us from the last arrival to the poll that
// sees it, more than a phase's matrix work; flags 0.33 us.  want[k]: phases every member must have completed with team k (0 = none);
// true = timed out.
__device__ __forceinline__ bool xwait(XCtx& c, const XcdArgs& a, const unsigned (&want)[4], FQL_GAS unsigned* err, unsigned* misc) {
#ifdef FQL_XSTAMPS
    const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (c.wave == 0) {
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (c.lane < XCD_NMEM) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (want[k]) ok = ok && __builtin_bit_cast(unsigned, ldx1(reinterpret_cast<const float*>(a.sync + 32 * (8 * k + c.g)), (unsigned)c.lane)) >= want[k];
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (c.lane == 0) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); misc[8 + c.team] = 1u; }
                break;
            }
        }
    }
#ifdef FQL_XSTAMPS
    const unsigned long long tw1 = __builtin_amdgcn_s_memrealtime();
#endif
    xsync(c);
#ifdef FQL_XSTAMPS
    c.tw[0] += tw1 - tw0; c.tw[1] += __builtin_amdgcn_s_memrealtime() - tw1; c.tw[4] += 1;
#endif
    return __builtin_amdgcn_readfirstlane((int)misc[8 + c.team]) != 0;   // (uniform over the team: every wave reads the same LDS word behind the barrier)
}
// done: the number of phases this member has now completed with its team
__device__ __forceinline__ void xarrive(XCtx& c, const XcdArgs& a, unsigned done) {
#ifdef FQL_XSTAMPS
    const unsigned long long ta0 = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this phase has reached the L2 ...
#ifdef FQL_XSTAMPS
    const unsigned long long ta1 = __builtin_amdgcn_s_memrealtime();
#endif
    xsync(c);
#ifdef FQL_XSTAMPS
    c.tw[2] += ta1 - ta0; c.tw[3] += __builtin_amdgcn_s_memrealtime() - ta1;
#endif
    if (c.tid == 0) stg(reinterpret_cast<float*>(a.sync + 32 * (8 * c.team + c.g)) + c.member, __builtin_bit_cast(float, done));   // ... before this member counts as arrived
}

// ---- the Euler chain (agents/fql.py:155-171) on its own path: flow_steps x (layer 0, hidden layers 1 .. nh - 1 with the head folded into the
// last).  Everything a phase needs but the activations is already on the CU - hidden kernels in LDS, the rank-16 update of layer 0 and the
// head rows in registers - so a phase is: wait, issue the panel loads, MFMA, reduce, epilogue, arrive; no op descriptor to decode.  A chain
// team walks its row tiles (t_first, t_first + t_step, ...) one per pass.
__device__ __forceinline__ void xchain_layer(XCtx& c, const XcdArgs& a, int l, const float (&w4f)[4]) {
    const int H = a.H, J = H >> 4, nh = a.chain_nl + 1;
    const float* A = a.hc[(l - 1) & 1];
    float* Co = a.hc[l & 1];
    const f32x4* wl = c.wlds + (size_t)(l - 1) * J * 64;
    const f32x4 bv = ldg4(a.chain_b[l - 1] + 16 * c.member + 4 * c.q);
    for (int t = c.t_first; t < c.RT; t += c.t_step) {
        const int rb = c.g * c.R + 16 * t;
        f32x4 av[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = c.wave + 4 * jj;
            av[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < J) av[jj] = ldx4(A, xtoff(c, rb, j, J));
        }
        f32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};   // two accumulators: the dependent-issue latency of one chain is 40 cycles against 32 of issue
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = c.wave + 4 * jj;
            const f32x4 wv = j < J ? wl[j * 64 + c.lane] : f32x4{0.f, 0.f, 0.f, 0.f};
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], av[jj][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], av[jj][1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], av[jj][2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], av[jj][3], acc1, 0, 0, 0);
        }
        xsync(c);
        c.red[c.wave * 64 + c.lane] = acc0 + acc1;
        xsync(c);
        if (c.wave == 0) {
            f32x4 v = c.red[c.lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += c.red[w * 64 + c.lane];
            v += bv;
            f32x4 g;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) g[tt] = gelu_f(v[tt]);
            if (l < nh - 1) stg4(Co + xtoff(c, rb, c.member, J), g);
            else {   // last hidden layer: this member's 16 columns times its 16 rows of the action head -> a partial of the velocity
                f32x4 pv{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) pv = __builtin_amdgcn_mfma_f32_16x16x4f32(w4f[tt], g[tt], pv, 0, 0, 0);
                if (4 * c.q < a.ad) stg4(a.vp + ((size_t)c.member * a.B + rb + c.r) * 16 + 4 * c.q, pv);
            }
        }
    }
}
// returns true when a wait timed out.  p counts this team's phases; team 0 enters at p0 = a.chain_p0 behind C0, team 1 at 0 waiting for team 0's C0.
__device__ __forceinline__ bool xchain_block(XCtx& c, const XcdArgs& a, unsigned p, FQL_GAS unsigned* err, unsigned* misc, const float (&w4f)[4]) {
    const int H = a.H, J = H >> 4, nh = a.chain_nl + 1;
    const bool active = 16 * c.member < H;
    f32x4 w0f;   // rows od .. od + 15 of the first kernel (the action block, t, zero padding) for this member's 16 columns
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = a.od + 4 * c.q + t;
        w0f[t] = (active && k < a.in_p) ? ldg(a.w0 + (size_t)k * H + 16 * c.member + c.r) : 0.f;
    }
    unsigned want[4] = {0u, 0u, 0u, 0u};
    want[0] = (unsigned)a.chain_p0;   // C0 is complete (team 0: its own previous phase)
    if (a.chain_p0 > 0 && xwait(c, a, want, err, misc)) return true;
    want[0] = 0u;
    xchain_fold(c, a, 0, c.t_first, c.t_step);
    for (int s = 0; s < a.fs; ++s) {
        if (s > 0) {
            want[c.team] = p;
            if (xwait(c, a, want, err, misc)) return true;
#ifdef FQL_XSTAMPS
            if (c.tid == 0) a.stamps[(((size_t)blockIdx.x * 4 + c.team) * a.stamp_stride + p) * 4] = __builtin_amdgcn_s_memrealtime();
#endif
            xchain_fold(c, a, s, c.t_first, c.t_step);
        }
        // layer 0: GELU(C0 + [a_s | t_s] W0[act rows, t row]); K = 16: one wave finishes a row tile by itself
        if (active && c.wave == 0)
            for (int t = c.t_first; t < c.RT; t += c.t_step) {
                const int rb = c.g * c.R + 16 * t;
                const f32x4 c0v = ldx4(a.c0, xtoff(c, rb, c.member, J));
                const f32x4 av = *reinterpret_cast<const f32x4*>(c.alds + (16 * t + c.r) * 16 + 4 * c.q);
                f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0f[tt], av[tt], acc, 0, 0, 0);
                acc += c0v;
                f32x4 g;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) g[tt] = gelu_f(acc[tt]);
                stg4(a.hc[0] + xtoff(c, rb, c.member, J), g);
            }
#ifdef FQL_XSTAMPS
#define XCS(k) do { if (c.tid == 0) a.stamps[(((size_t)blockIdx.x * 4 + c.team) * a.stamp_stride + p) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define XCS(k) do {} while (0)
#endif
        XCS(1);
        xarrive(c, a, p + 1u); XCS(2); ++p;
        for (int l = 1; l < nh; ++l) {
            want[c.team] = p;
            if (xwait(c, a, want, err, misc)) return true;
            XCS(0);
            if (active) xchain_layer(c, a, l, w4f);
            XCS(1);
            xarrive(c, a, p + 1u); XCS(2); ++p;
        }
    }
    return false;
}

// 1024 threads: four teams of four waves (see XPhase).
__global__ __launch_bounds__(1024, 4) void fql_xcd_kernel(const XcdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xlds[];
    XCtx c;
    c.lane = threadIdx.x & 63;
    const int wave16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.team = wave16 >> 2; c.wave = wave16 & 3; c.tid = threadIdx.x & 255;
    c.r = c.lane & 15; c.q = c.lane >> 4;
    c.R = a.R; c.RT = a.RT; c.st2 = nullptr; c.bt = 0u;
    for (int i = 0; i < 6; ++i) c.tw[i] = 0;
    c.g = (int)(__builtin_amdgcn_s_getreg(6164) & 7u);   // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
    float* tl = xlds + c.team * (8 * 64 * 4 + 256);               // per team: reduction slots [4 waves][2 tiles][64] float4, statistics [4][2][16][2]
    c.red = reinterpret_cast<f32x4*>(tl);
    c.stat = tl + 8 * 64 * 4;
    float* sh = xlds + 4 * (8 * 64 * 4 + 256);
    c.alds = sh;                                                  // [R][16] (chain teams)
    unsigned* misc = reinterpret_cast<unsigned*>(sh + XCD_MAXROWS * 16);   // 16 words: ticket, team barrier words [4..8), time-out flags [8..12)
    f32x4* wl = reinterpret_cast<f32x4*>(sh + XCD_MAXROWS * 16 + 16);
    c.wlds = a.chain_nl > 0 ? wl : nullptr;
    c.stl = reinterpret_cast<unsigned long long*>(sh + XCD_MAXROWS * 16 + 16 + a.chain_nl * (a.H >> 4) * 64 * 4) + 16 * c.team;
    c.bar = (__attribute__((address_space(3))) unsigned*)(misc + 4 + c.team);
    c.t_first = 0; c.t_step = 1;
    FQL_GAS unsigned* err = (FQL_GAS unsigned*)(a.sync + 32 * 40);
    if (threadIdx.x == 0) {
        misc[0] = __hip_atomic_fetch_add((FQL_GAS unsigned*)(a.sync + 32 * (32 + c.g)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 4; i < 12; ++i) misc[i] = 0u;
    }
    {   // pull the op and phase tables into this XCD's L2 (they are read with scalar loads, one cold miss each otherwise)
        const int nphase = a.nphase0 + a.nphase1;
        const __attribute__((address_space(4))) int* l0 = (const __attribute__((address_space(4))) int*)(a.phases + (a.nphase0 - 1));
        const __attribute__((address_space(4))) int* l1 = (const __attribute__((address_space(4))) int*)(a.phases + (nphase - 1));
        const int nops = max(l0[0] + l0[1], l1[0] + l1[1]);
        float acc = 0.f;
        for (int i = threadIdx.x * 16; i < (int)(nops * sizeof(XOp) / 4); i += 1024 * 16) acc += ldg(reinterpret_cast<const float*>(a.ops) + i);
        for (int i = threadIdx.x * 16; i < (int)(nphase * sizeof(XPhase) / 4); i += 1024 * 16) acc += ldg(reinterpret_cast<const float*>(a.phases) + i);
        if (acc == 1.2345e-30f) misc[15] = 1u;   // (keeps the loads alive)
    }
    __syncthreads();
    c.member = __builtin_amdgcn_readfirstlane((int)misc[0]);   // (an LDS read: uniform, but only this tells the compiler)
    if (c.member >= XCD_NMEM) {   // more than 32 workgroups on this XCD: not a placement this kernel runs on
        if (threadIdx.x == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // the Euler chain's hidden kernels: this member's 16 output columns of each, fragment-major, resident for the whole launch
    const int JH = a.H >> 4;
    if (c.wlds && 16 * c.member < a.H) {
        for (int l = 0; l < a.chain_nl; ++l)
            for (int i = threadIdx.x; i < JH * 64; i += 1024) {
                const int j = i >> 6, ln = i & 63, rr = ln & 15, qq = ln >> 4;
                f32x4 v;
                const float* p = a.chain_w[l] + (size_t)(16 * j + 4 * qq) * a.H + 16 * c.member + rr;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = ldg(p + (size_t)t * a.H);
                wl[((size_t)l * JH + j) * 64 + ln] = v;
            }
    }
    // this member's 16 rows of the action head as the first MFMA operand of the partial product: lane (n, q) holds W4[16 member + 4 q + t][n]
    float w4f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = 16 * c.member + 4 * c.q + t;
        w4f[t] = (a.w4 && k < a.H && c.r < a.ap) ? ldg(a.w4 + (size_t)k * a.ap + c.r) : 0.f;
    }
    __syncthreads();   // (the last workgroup-wide barrier: from here on the teams go their own ways)
    if (c.team == 1) {   // the second chain pipeline: nothing but the chain block
        if (a.nct == 2 && c.wlds && a.skip_team != 1) { c.t_first = 1; c.t_step = 2; xchain_block(c, a, 0u, err, misc, w4f); }
        return;
    }
    const bool filler = c.team >= 2;
    const int np = filler ? a.nphase1 : a.nphase0;
    const XPhase* phs = a.phases + (filler ? a.nphase0 : 0);
    for (int p = 0; p < np; ++p) {
        XPhase ph;
        {
            const __attribute__((address_space(4))) int* src = (const __attribute__((address_space(4))) int*)(phs + p);
            ph.first = src[0]; ph.count = src[1]; ph.count_a = src[2]; ph.chain = src[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) ph.wait[k] = src[4 + k];
        }
        if (ph.chain > 0 && c.wlds && c.team != a.skip_team) {   // the Euler chain: `chain` phases on the dedicated path
            if (a.nct == 2) { c.t_first = 0; c.t_step = 2; }
            if (xchain_block(c, a, (unsigned)p, err, misc, w4f)) break;
            c.t_first = 0; c.t_step = 1;
            p += ph.chain - 1;
            continue;
        }
        unsigned want[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) want[k] = (unsigned)ph.wait[k];
        if (filler) { want[2] = max(want[2], (unsigned)p); want[3] = max(want[3], (unsigned)p); }
        else want[0] = max(want[0], (unsigned)p);
        if ((want[0] | want[1] | want[2] | want[3]) && xwait(c, a, want, err, misc)) break;
#ifdef FQL_XSTAMPS
        unsigned long long* stp = a.stamps + (((size_t)blockIdx.x * 4 + c.team) * a.stamp_stride + p) * 4;
        if (c.tid == 0) stp[0] = __builtin_amdgcn_s_memrealtime();
        c.st2 = (p == a.stamp_phase && c.team == a.stamp_team) ? a.stamps2 + (size_t)blockIdx.x * 16 : nullptr;
        XST(c, 8);
#endif
        if (c.team != a.skip_team) {
            const int o0 = c.team == 3 ? ph.first + ph.count_a : ph.first;
            const int o1 = c.team == 2 ? ph.first + ph.count_a : ph.first + ph.count;
            for (int oi = o0; oi < o1; ++oi) xrun(c, a, a.ops + oi, w4f);
        }
        XST(c, 7);
#ifdef FQL_XSTAMPS
        if (c.tid == 0) stp[1] = __builtin_amdgcn_s_memrealtime();
        if (c.st2 && c.tid == 0) for (int k = 0; k < 16; ++k) c.st2[k] = c.stl[k];
#endif
        xarrive(c, a, (unsigned)p + 1u);
#ifdef FQL_XSTAMPS
        if (c.tid == 0) { stp[2] = __builtin_amdgcn_s_memrealtime(); stp[3] = __builtin_amdgcn_s_memtime(); if (p == 0) a.stamps2[256 * 16 + blockIdx.x] = (unsigned long long)((c.g << 8) | c.member); }
#endif
    }
#ifdef FQL_XSTAMPS
    if (c.tid == 0 && c.team == 0) for (int i = 0; i < 5; ++i) a.stamps2[256 * 17 + blockIdx.x * 8 + i] = c.tw[i];
#endif
}

#define FQL_XCD_LDS_FLOATS(chain_nl, H) (4 * (8 * 64 * 4 + 256) + XCD_MAXROWS * 16 + 16 + (chain_nl) * ((H) / 16) * 64 * 4 + 128)
