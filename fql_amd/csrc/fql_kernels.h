// Device code of the FQL step engine, gfx950 (MI355X / CDNA4) only.
//
// All dense contractions run on the fp32-input matrix cores (v_mfma_f32_16x16x4_f32): exact fp32
// products, fp32 accumulate == an fmaf chain, which is what keeps the step inside the fp32 parity
// tolerance against the CPU oracle.  Feature dimensions are padded to multiples of 16 in HBM (zero
// padded weights / activations), the batch dimension to multiples of 16.
//
// Reference op sites each kernel replaces are cited as file:line of zhouzypaul/fql.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers read out of a task table are generic to the compiler and would become flat_load/flat_store
// (which also tick lgkmcnt and serialise against LDS traffic).  Everything the engine hands to a kernel is
// device global memory, so say so.
#define FQL_GAS __attribute__((address_space(1)))
__device__ __forceinline__ float ldg(const float* p) { return *(const FQL_GAS float*)p; }
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const FQL_GAS f32x4*)p; }
__device__ __forceinline__ void stg(float* p, float v) { *(FQL_GAS float*)p = v; }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *(FQL_GAS f32x4*)p = v; }

#define FQL_THREADS 256

// Division of a small non-negative int by a kernel-uniform divisor without the ~40-instruction integer division sequence (100+ for 64-bit indices):
// q = umulhi(x, ceil(2^32 / d)), exact whenever x d < 2^32.  The convolution / pooling kernels index pixels through W, W + 2, channel quads ...: their
// epilogues and staging loops were bound by these divisions (experiments/conv_bench_st: 4.1 us of staging per 7.8 us workgroup against 1.65 us of MFMA).
struct FastDiv {
    unsigned m;   // 0: plain division (d == 1, or the caller's largest dividend is out of the exact range)
    int d;
    __device__ __forceinline__ explicit FastDiv(int dd) : m(dd > 1 ? 0xFFFFFFFFu / (unsigned)dd + 1u : 0u), d(dd) {}
    // for dividends that grow with the batch (flat element indices): falls back to the plain division when max_x d >= 2^32
    __device__ __forceinline__ FastDiv(int dd, unsigned long long max_x) : m(dd > 1 && max_x * (unsigned long long)dd < (1ull << 32) ? 0xFFFFFFFFu / (unsigned)dd + 1u : 0u), d(dd) {}
    __device__ __forceinline__ int div(int x) const { return m ? (int)__umulhi((unsigned)x, m) : (d == 1 ? x : (int)((unsigned)x / (unsigned)d)); }
    __device__ __forceinline__ void divmod(int x, int& q, int& r) const { q = div(x); r = x - q * d; }
};

// x / d for small indices through ONE v_rcp_f32 (rd = frcp(d), computed once per divisor): floor((x + 0.5) rd).  (x + 0.5) / d is at least 0.5 / d away
// from an integer and the product is off by < (x / d) 2^-21, so the result is exact for x < 2^20 - every tile / element index here.  The integer
// division it replaces is ~40 instructions; the 16-row kernel's prologue alone had ~30 of them per thread.
__device__ __forceinline__ float frcp(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int sdiv(int x, float rd) { return (int)(((float)x + 0.5f) * rd); }

// XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, each with a private L2 that is cold after
// every kernel boundary, so workgroups b, b + 8, ... share an L2.  Handing each XCD a compact (ntm / gm) x (ntn / gn) block of a task's tile grid
// (gm gn = 8) instead of every 8th tile makes the tiles that share an A row panel or a B column panel fetch it from the Infinity Cache once per
// XCD that needs it (gn + gm panel fetches instead of 8 + 8).  `local` = tile index inside the task, whose first workgroup is a multiple of 8.
// Placement is a speed matter only: results do not depend on which workgroup computes which tile.
__device__ __forceinline__ void xcd_tile(int local, int ntm, int ntn, int gm, int& tm, int& tn) {
    const int gs = 31 - __clz(gm);                      // gm is 1, 2, 4 or 8: every division below is a shift but one
    const int x = local & 7, j = local >> 3;
    const int bn = ntn >> (3 - gs), bm = ntm >> gs;     // ntn / gn, ntm / gm (exact: checked where the order is chosen)
    const int xm = x >> (3 - gs);                       // x / gn
    const int jm = sdiv(j, frcp(bn));
    tm = xm * bm + jm;
    tn = (x - (xm << (3 - gs))) * bn + (j - jm * bn);
}

// ------------------------------------------------------------------------------------------------
// precision = 2 ("bf16x3", SURVEY 8b config key `precision`): an fp32 operand x is split into two bf16 values
//   hi = bf16(x), lo = bf16(x - hi)            (both round-to-nearest-even; x - hi is exact in fp32)
// and a product a b is evaluated as a_hi b_hi + (a_hi b_lo + a_lo b_hi) on the bf16 matrix cores
// (v_mfma_f32_16x16x32_bf16, fp32 accumulation; the two small terms on an accumulator of their own).  What is dropped is
// a_lo b_lo and the split's own residual: <= 2^-16 |a b| per product, ~2^-18 typical with signs that average out
// (fp32 itself: 2^-24).  16 cycles per 16x16x32 MFMA against 32 per 16x16x4 fp32 MFMA: 3 x 16 cycles do the work of 8 x 32.
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// {bf16(x0) in bits 0-15, bf16(x1) in bits 16-31}: one v_cvt_pk_bf16_f32.  Through the compiler's own conversion, NOT inline
// assembly: the packed subtraction in front of the second conversion (v_pk_add_f32) needs a wait state before its result is read,
// which the hazard recognizer only inserts for instructions it can see (an inline-asm conversion read garbage now and then).
__device__ __forceinline__ unsigned cvt_pk_bf16(float x0, float x1) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
}
__device__ __forceinline__ void bsplit2(float x0, float x1, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    lo = cvt_pk_bf16(x0 - h0, x1 - h1);
}
__device__ __forceinline__ void bsplit4(const f32x4& v, u32x2& hi, u32x2& lo) {
    unsigned h0, l0, h1, l1;
    bsplit2(v[0], v[1], h0, l0);
    bsplit2(v[2], v[3], h1, l1);
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_bf16_k16(const u32x2& a, const u32x2& b, const f32x4& acc) {   // v_mfma_f32_16x16x16_bf16: lane (r, q) owns k = 4q .. 4q + 3
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc, 0, 0, 0);
}
__device__ __forceinline__ u32x4 ldg4u(const unsigned* p) { return *(const FQL_GAS u32x4*)p; }
// transposed LDS read (ds_read_b64_tr_b16): within each group of 16 lanes, lane 4r + p supplies the address of row r, columns
// 4p .. 4p+3 of a 4 x 16 block of 16-bit elements; lane i receives column i (element e = row e).  EXEC must be all ones.
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned* p) {
    typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)p));
}

// Diagnostics build (-DFQL_TIMELINE): every instrumented launch records the wall-clock (100 MHz) time of its first and last
// workgroup entry and of its last exit, so the REAL overlapped schedule of the two lanes can be read back (rocprofv3 serialises
// the graph's branches).  In the product build the marks compile to nothing; the launch id argument stays for one code path.
#ifdef FQL_TIMELINE
#define FQL_TL_MAX 128       // launches
#define FQL_TL_WGS 2048      // workgroups per launch (those beyond are not recorded)
__device__ unsigned long long g_fql_tl[FQL_TL_MAX][FQL_TL_WGS][2];   // [launch id][workgroup]: entry, exit (own slot: plain stores, no atomics)
__device__ __forceinline__ void tl_enter(int id) {
    if (id >= 0 && id < FQL_TL_MAX && threadIdx.x == 0 && blockIdx.x < FQL_TL_WGS) g_fql_tl[id][blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void tl_exit(int id) {
    if (id >= 0 && id < FQL_TL_MAX && threadIdx.x == 0 && blockIdx.x < FQL_TL_WGS) g_fql_tl[id][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
}
#else
#define tl_enter(id) ((void)0)
#define tl_exit(id) ((void)0)
#endif

// ------------------------------------------------------------------------------------------------
// task descriptors (built once on the host, read from HBM by every workgroup of a launch)
// ------------------------------------------------------------------------------------------------
enum : int {
    GF_TRANS_B = 1 << 0,    // B operand is W^T (dgrad):   C = A * W^T,  W stored [K_out=N][..]
    GF_BIAS = 1 << 1,       // + bias[n]
    GF_GELU = 1 << 2,       // GELU-tanh epilogue                      utils/networks.py:46,56
    GF_SAVE_Z = 1 << 3,     // store GELU'(pre-activation) beside the GELU output: the backward pass then needs no transcendental
                            // (LayerNorm backward reads g and g', the un-normalised dgrad epilogue multiplies by g')
    GF_A_LN = 1 << 4,       // LayerNorm the A tile in LDS before the product   utils/networks.py:58
    GF_LN_WRITE = 1 << 5,   // column-tile 0 also stores LN(A) and (mean, rstd) for backward
    GF_GELUGRAD = 1 << 6,   // epilogue: C = acc * Zprev, Zprev = the stored GELU'(z)  (dgrad through an un-normalised layer)
    GF_EULER = 1 << 7,      // epilogue: a += v / flow_steps, t column := t_next    agents/fql.py:166-169
    GF_EULER_LAST = 1 << 8, // ... and store clip(a, -1, 1) as the distillation target  agents/fql.py:170
    GF_CLIP_OUT = 1 << 9,   // epilogue: C = clip(acc + bias, -1, 1)            agents/fql.py:152
    GF_A_EULER0 = 1 << 11,  // fused Euler step, part 1: A tile = GELU(C0 + a W0[act rows] + t W0[t row]) built in LDS
    GF_HEAD_PART = 1 << 12, // fused Euler step, part 3: epilogue multiplies the GELU tile into the action head (partials)
    GF_OS_SCATTER = 1 << 13, // one-step head on [next_obs; obs; obs] rows: also write clip(out) into the critic inputs
    GF_A_LOSSACT = 1 << 17,  // A tile = d(actor loss)/d(one-step actions) built in the prologue (agents/fql.py:66-79): no loss kernel on the critical path
    GF_A_EULFIN = 1 << 15,   // with GF_A_LOSSACT: the distillation target is finished in the same prologue - clip(a_9 + (sum of the last step's head
                             // partials + head bias) / flow_steps), agents/fql.py:169-170 - instead of by fql_euler_finish_kernel one launch earlier:
                             // aux = a of the last step [M, i0], aux2 = partials [ln_width][M][i0], eb = head bias, f1 = 1 / flow_steps; evp = target OUT (column tile 0)
    GF_RELUGRAD = 1 << 14,   // epilogue: C = (Zprev > 0) ? acc : 0  (dgrad through the encoder's final ReLU, utils/encoders.py:92)
    GF_LN_PART = 1 << 10,   // gemm64 epilogue: per-row (sum, sum sq) of this 64-column tile -> aux[row][i1 tiles][2]
    GF_C_FRAGT = 1 << 16,   // gemm16 epilogue: C (+ bias) in the layout a TRANSPOSED 16 x 16 accumulator tile reads as one dwordx4 per lane, 1 KB contiguous
                            // per tile: [M/16][N/16][q = (n % 16) / 4][c = m % 16][n % 4] (fql_chain_split_kernel variant A: lane (c, q) owns row c, columns 4 q ..)
    GF_C_FRAG = 1 << 18,    // gemm16 epilogue: C (+ bias) stored in accumulator-fragment-major layout [M/4][N][4] (one dwordx4 per lane;
                            // read back as one dwordx4 per MFMA tile by fql_chain_kernel variant A)
};

struct GemmTask {
    const float* A;     // [M, K] lda
    const float* B;     // W: [K, N] ldb (or [N.., K..] read transposed with GF_TRANS_B)
    const float* bias;  // [N]
    float* C;           // [M, N] ldc
    float* Zout;        // [M, N] ldc   (GF_SAVE_Z): GELU'(z)
    const float* Zprev; // [M, N] ldc   (GF_GELUGRAD: stored GELU'(z) of the previous layer; GF_RELUGRAD: its output)
    const float* ln_g;  // [K]          (GF_A_LN)
    const float* ln_b;  // [K]
    float* ln_xout;     // [M, K] lda   (GF_LN_WRITE)
    float* ln_stats;    // [M, 2]       (GF_LN_WRITE)
    float* aux;         // GF_EULER: X_eu [M, ld = i0];  gemm64 GF_LN_PART: partial sums out [M][i1][2]
    float* aux2;        // GF_EULER_LAST: target actions [M, N];  gemm64 GF_A_LN: partial sums in [M][i0][2]
    int M, N, K;
    int lda, ldb, ldc;
    int ln_width;       // real (unpadded) width for LN statistics
    int flags;
    int tile0;          // first workgroup index of this task inside the launch
    int ntn;            // workgroup tiles along N
    int wk;             // K-split ways among the 4 waves (1, 2 or 4); column tiles per workgroup = 4 / wk
    int tmt;            // 16-row tiles per workgroup (1 or 2)
    int i0, i1, i2;     // GF_EULER: ld of aux, column offset of the action block, act_dim
    float f0, f1;       // GF_EULER: 1/flow_steps, t_next;  GF_A_EULER0: 1/flow_steps, t of this step
    // fused Euler step (agents/fql.py:166-169 with layers 0+1 and last-hidden+head each in one launch)
    const float* ea_in;  // [M, i0] current actions a_s (row stride i0)
    float* ea_out;       // [M, ap] a_s as used by this step (written by column tile 0) or null
    const float* ew;     // A_EULER0: W0 rows of the action block and t, [ad+1][K]
    const float* ew4;    // HEAD_PART: head kernel [N][ap]
    float* evp;          // head partials [ntiles][M][ap]: read by A_EULER0 (null on step 0), written by HEAD_PART
    const float* eb;     // head bias [ap]
    int e_ntp;           // number of partial tiles to fold (A_EULER0)
    int xg;              // 32-row tile bodies: XCD-aware tile order with gm = xg row groups (xcd_tile); 0 = row-major tile order
#ifdef FQL_STAMPS
    unsigned long long* stamps64;   // diagnostics build: gemm64 phase stamps [workgroup][8]
#endif
};

struct WgradTask {
    const float* X;   // [M, Kin] ldx  (layer input)
    const float* dZ;  // [M, N]   ldz
    float* dW;        // [Kin, N] ldw
    float* db;        // [N] or null
    int M, Kin, N;
    int ldx, ldz, ldw;
    int tile0, ntn;
};

struct LnBwdTask {
    const float* dY;     // [M, H] ld   gradient w.r.t. LN output
    const float* Z;      // [M, H] ld   GELU'(z) of the layer as stored by the forward pass
    const float* Gv;     // [M, H] ld   GELU(z) = what the LayerNorm normalised
    const float* stats;  // [M, 2] mean, rstd
    const float* gamma;  // [H]
    float* dZ;           // [M, H] ld   out: gradient w.r.t. the pre-activation
    float* dgamma;       // [H] or null (param grads; column sums over the batch)
    float* dbeta;        // [H] or null
    int M, H, ld, width;
    int tile0, ntiles_rows; // row tasks first, then column-sum tiles
    // scalar head (critic Q): dY[m][k] = dq[m * ldq] * wq[k * ldw] is a rank-1 product, no dgrad GEMM needed
    const float* dq;
    const float* wq;
    int ldq, ldw;
};

// ------------------------------------------------------------------------------------------------
// math helpers
// ------------------------------------------------------------------------------------------------
// tanh via one v_exp_f32 and one reciprocal: tanh(u) = sign(u) (1 - e) / (1 + e), e = exp(-2|u|) in (0, 1].
// Absolute error ~1e-7 (an ulp of 1), which is what GELU / GELU' see; libm tanhf costs ~5x the instructions and
// sat on the critical path of every layer epilogue.
__device__ __forceinline__ float fast_tanh(float u) {
    const float e = __expf(-2.0f * fabsf(u));
    const float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, u);
}
#ifdef FQL_GELU_TANH   // the round-1 form (kept for A/B measurements)
__device__ __forceinline__ float gelu_f(float x) {
    // flax nn.gelu (approximate=True): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + fast_tanh(u));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float th = fast_tanh(u);
    const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * du;
}
#else
// flax nn.gelu (approximate=True) = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)   (utils/networks.py:46).
// 0.5 (1 + tanh u) = sigmoid(2u) = 1 / (1 + exp(-2u)), so GELU = x / (1 + e) with e = exp2(x (C0 + C1 x^2)),
// C0 = -2 sqrt(2/pi) log2(e), C1 = 0.044715 C0: five VALU ops + v_exp_f32 + v_rcp_f32 (the tanh form cost ~15 + 2; this sits in
// every layer epilogue and, 16 x 512 times per workgroup, in the first launch of every Euler step).  e is clamped at 1e30 so
// that e / (1 + e) stays finite for very negative x (GELU -> x * 1e-30 ~ -0).
#define FQL_GELU_C0 (-2.302208198f)
#define FQL_GELU_C1 (-0.1029432427f)
__device__ __forceinline__ void gelu_core(float x, float& r, float& e, float& x2) {
    x2 = x * x;
    e = fminf(__builtin_amdgcn_exp2f(x * fmaf(FQL_GELU_C1, x2, FQL_GELU_C0)), 1e30f);
    r = __builtin_amdgcn_rcpf(1.0f + e);   // sigmoid(2u)
}
__device__ __forceinline__ float gelu_f(float x) {
    float r, e, x2;
    gelu_core(x, r, e, x2);
    return x * r;
}
// d/dx [x s(2u)] = s + x s (1 - s) (2u)',  (2u)' = 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2),  1 - s = e r
__device__ __forceinline__ float gelu_grad_f(float x) {
    float r, e, x2;
    gelu_core(x, r, e, x2);
    return r * fmaf(x * (e * r), fmaf(0.2140644f, x2, 1.5957691216f), 1.0f);
}
#endif
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
#ifdef FQL_GELU_TANH
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float th = fast_tanh(u);
    const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
    g = 0.5f * x * (1.0f + th);
    dg = 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * du;
#else
    float r, e, x2;
    gelu_core(x, r, e, x2);
    g = x * r;
    dg = r * fmaf(x * (e * r), fmaf(0.2140644f, x2, 1.5957691216f), 1.0f);
#endif
}
__device__ __forceinline__ float clip1(float x) { return fminf(fmaxf(x, -1.0f), 1.0f); }

// Philox4x32-10 (Salmon et al. 2011), counter-based: the engine's own RNG stream (the reference's
// threefry stream is not reproducible here).
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // [0, 1)
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// element `col` of a standard-normal row vector identified by (key, step, tensor, row)
__device__ __forceinline__ float rng_normal(uint64_t key, uint64_t step, uint32_t tensor, uint32_t row,
                                            uint32_t col) {
    uint32_t r[4];
    philox4x32((uint32_t)step, (uint32_t)(step >> 32) ^ (tensor << 24), row, col >> 2, (uint32_t)key,
               (uint32_t)(key >> 32), r);
    // two Box-Muller pairs per Philox block: (r0, r1) -> cols 4j, 4j+1 ; (r2, r3) -> 4j+2, 4j+3
    const int pair = (col >> 1) & 1;
    const float u1 = ((float)(r[2 * pair] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = u32_to_unit(r[2 * pair + 1]);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    return (col & 1) ? rad * s : rad * c;
}
__device__ __forceinline__ float rng_uniform(uint64_t key, uint64_t step, uint32_t tensor, uint32_t row) {
    uint32_t r[4];
    philox4x32((uint32_t)step, (uint32_t)(step >> 32) ^ (tensor << 24), row, 0u, (uint32_t)key,
               (uint32_t)(key >> 32), r);
    return u32_to_unit(r[0]);
}
__device__ __forceinline__ uint32_t rng_u32(uint64_t key, uint64_t step, uint32_t tensor, uint32_t row) {
    uint32_t r[4];
    philox4x32((uint32_t)step, (uint32_t)(step >> 32) ^ (tensor << 24), row, 0u, (uint32_t)key,
               (uint32_t)(key >> 32), r);
    return r[1];
}

template <typename T>
__device__ __forceinline__ int find_task(const T* tasks, int ntasks, int b) {
    int t = 0;
    for (int i = 1; i < ntasks; ++i)
        if (b >= tasks[i].tile0) t = i;
    return t;
}

// ------------------------------------------------------------------------------------------------
// K1-K6: Dense (+bias) (+GELU) with optional LayerNorm prologue on the A tile; also dgrad (C = A W^T)
//   utils/networks.py:53-58 (Dense -> GELU -> LN), agents/fql.py:166-170 (Euler epilogue)
//
// Workgroup = 4 waves on a 16-row tile.  The 4 waves are arranged as NW column tiles x WK K-parts
// (NW * WK = 4): wide layers use 2 x 2 (16 x 32 output tile, each wave half of K), 16-column layers
// (the action / Q heads) 1 x 4.  That puts a [256 x 512] x [512 x 512] layer of the sequential Euler
// chain on 256 workgroups x 4 waves = every SIMD of the chip with 64 MFMAs each; the partial
// accumulators meet in LDS.  The A tile [16 x K] is staged in LDS (row stride K+4 floats) by coalesced
// 16-byte loads; with GF_A_LN the 16 rows are normalised in place first (the row statistics need the
// whole row = the K extent of the tile).  B fragments are prefetched from L2 into registers one
// 128-deep K chunk (32 VGPRs) ahead of the MFMAs that consume them: at one wave per SIMD nothing
// else hides the load latency.  K order inside the fma chain: lane (r, q) takes k = 16 g + 4 q + s for
// MFMA step s of group g, so its A fragment is one ds_read_b128.
// ------------------------------------------------------------------------------------------------
template <bool TRANS>
__device__ __forceinline__ void gemm_load_chunk(float (&b)[32], const float* __restrict__ bp, size_t ldb) {
    if (TRANS) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 v = ldg4(bp + 16 * g);
            b[4 * g] = v[0]; b[4 * g + 1] = v[1]; b[4 * g + 2] = v[2]; b[4 * g + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) b[4 * g + s] = ldg(bp + (size_t)(16 * g + s) * ldb);
    }
}
// Two accumulator chains (even / odd k-groups): v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent
// accumulate needs 40, so a single chain runs at 80 % of the matrix rate.
__device__ __forceinline__ void gemm_mma_chunk(f32x4& acc, f32x4& acc2, const float (&b)[32], const float* arow) {
#pragma unroll
    for (int g = 0; g < 8; g += 2) {
        const float4 a = *reinterpret_cast<const float4*>(arow + 16 * g);
        const float4 a2 = *reinterpret_cast<const float4*>(arow + 16 * g + 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[4 * g], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b[4 * g + 4], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[4 * g + 1], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b[4 * g + 5], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[4 * g + 2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, b[4 * g + 6], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[4 * g + 3], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, b[4 * g + 7], acc2, 0, 0, 0);
    }
}

// B-operand addressing of one wave: element (k, n0 + c), k = 16 g + 4 q (+ s)
struct BAddr {
    const float* base;
    size_t ldb, kstep;
};
template <bool TRANS>
__device__ __forceinline__ BAddr gemm_baddr(const GemmTask& T, int n0, int c, int q) {
    BAddr a;
    a.ldb = T.ldb;
    a.base = TRANS ? T.B + (size_t)(n0 + c) * a.ldb + 4 * q : T.B + (size_t)(4 * q) * a.ldb + n0 + c;
    a.kstep = TRANS ? 1 : a.ldb;
    return a;
}
// b0 / b1 already hold chunks 0 / 1 of [gbeg, gend) when they exist (issued before the A tile landed).
// TMT row tiles (16 rows each) share every B fragment: TMT x the MFMA work per loaded B byte.
template <bool TRANS, int TMT>
__device__ __forceinline__ void gemm_wave(f32x4 (&acc)[TMT], const BAddr& ba, const float* lds_a, int S, int gbeg, int gend,
                                          int c, int q, float (&b0)[32], float (&b1)[32]) {
    const float* __restrict__ arow = lds_a + c * S + 4 * q;
    int g = gbeg;
    const int nfull = (gend - gbeg) >> 3;
    f32x4 acc2[TMT];
#pragma unroll
    for (int r = 0; r < TMT; ++r) acc2[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nfull; i += 2) {
#pragma unroll
        for (int r = 0; r < TMT; ++r) gemm_mma_chunk(acc[r], acc2[r], b0, arow + r * 16 * S + 16 * g);
        if (i + 2 < nfull) gemm_load_chunk<TRANS>(b0, ba.base + (size_t)(16 * (g + 16)) * ba.kstep, ba.ldb);
        if (i + 1 < nfull) {
#pragma unroll
            for (int r = 0; r < TMT; ++r) gemm_mma_chunk(acc[r], acc2[r], b1, arow + r * 16 * S + 16 * (g + 8));
            if (i + 3 < nfull) gemm_load_chunk<TRANS>(b1, ba.base + (size_t)(16 * (g + 24)) * ba.kstep, ba.ldb);
        }
        g += 16;
    }
#pragma unroll
    for (int r = 0; r < TMT; ++r) acc[r] += acc2[r];
    g = gbeg + 8 * nfull;
    // tail: < 8 groups, all loads first
    const int nt = gend - g;
    if (nt > 0) {
        float bt[28];
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            if (t < nt) {
                if (TRANS) {
                    const f32x4 v = ldg4(ba.base + (size_t)(16 * (g + t)));
                    bt[4 * t] = v[0]; bt[4 * t + 1] = v[1]; bt[4 * t + 2] = v[2]; bt[4 * t + 3] = v[3];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bt[4 * t + s] = ldg(ba.base + (size_t)(16 * (g + t) + s) * ba.ldb);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < TMT; ++r) {
                    const float4 a = *reinterpret_cast<const float4*>(arow + r * 16 * S + 16 * (g + t));
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bt[4 * t], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bt[4 * t + 1], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bt[4 * t + 2], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bt[4 * t + 3], acc[r], 0, 0, 0);
                }
            }
        }
    }
}

// NA = float4 loads per thread that cover 16 rows of the A tile (16 * K / 4 <= NA * 256);
// TMT = 16-row tiles per workgroup (1: latency lane, 2: throughput lane).
template <int NA, int TMT, bool EUL>
__device__ __forceinline__ void gemm16_body(const GemmTask& T, float* lds) {
    const int local = blockIdx.x - T.tile0;
    const int tm = sdiv(local, frcp(T.ntn)), tn = local - tm * T.ntn;
    const int row0 = tm * 16 * TMT;
    const int K = T.K, N = T.N;
    const int S = K + 4;  // LDS row stride (floats)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int flags = T.flags;
    const int WK = T.wk, wks = WK >> 1, NW = 4 >> wks;   // WK is 1, 2 or 4: wks = log2(WK)
    float* red = lds + 16 * TMT * S;  // [WK-1][NW][TMT][64] float4 partial accumulators
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define STAMP() do { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    STAMP();
#else
#define STAMP() do {} while (0)
#endif

    // ---- issue every independent load of the tile up front: the A tile (<= 16 x 16 B per thread and row
    // tile), the wave's first two B chunks and its bias.  At one wave per SIMD only explicit parallel issue
    // hides the ~0.3-1.5 us (cold, cross-XCD) load latency; a rolled staging loop would serialise it.
    const int nt = wave & (NW - 1), kp = wave >> (2 - wks);
    const int n0 = (tn * NW + nt) * 16;
    const int c = lane & 15, q = lane >> 4;
    const bool active = n0 < N;
    const int G = K >> 4;
    const int gbeg = (kp * G) >> wks, gend = ((kp + 1) * G) >> wks;
    const bool transb = (flags & GF_TRANS_B) != 0;
    const int k4 = K >> 2;
    const float rk4 = frcp(k4);
    const int nA = 16 * k4;
    float b0[32], b1[32];
    BAddr ba = transb ? gemm_baddr<true>(T, n0, c, q) : gemm_baddr<false>(T, n0, c, q);
    float bias = 0.f;
    float* ea = red + 1024 * TMT;  // [16][32] fused Euler: actions of this step
    float* hs = ea + 512;          // [16][36] fused Euler: last hidden tile feeding the head partial
    const bool euler0 = EUL && TMT == 1 && (flags & GF_A_EULER0);
    const bool head = EUL && TMT == 1 && (flags & GF_HEAD_PART);
    auto issue_b = [&]() {
#pragma unroll
        for (int i = 0; i < 32; ++i) { b0[i] = 0.f; b1[i] = 0.f; }  // defined on every path: stay in VGPRs
        if (active) {
            const int nfull = (gend - gbeg) >> 3;
            if (transb) {
                if (nfull > 0) gemm_load_chunk<true>(b0, ba.base + (size_t)(16 * gbeg) * ba.kstep, ba.ldb);
                if (nfull > 1) gemm_load_chunk<true>(b1, ba.base + (size_t)(16 * (gbeg + 8)) * ba.kstep, ba.ldb);
            } else {
                if (nfull > 0) gemm_load_chunk<false>(b0, ba.base + (size_t)(16 * gbeg) * ba.kstep, ba.ldb);
                if (nfull > 1) gemm_load_chunk<false>(b1, ba.base + (size_t)(16 * (gbeg + 8)) * ba.kstep, ba.ldb);
            }
            if ((flags & GF_BIAS) && kp == 0) bias = ldg(T.bias + n0 + c);
        }
    };
    float bw4[8];  // head-kernel fragments of this workgroup's 32 hidden columns (wave 0 only)
#pragma unroll
    for (int i = 0; i < 8; ++i) bw4[i] = 0.f;
    if (head && wave == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = tn * 32 + 16 * g + 4 * q + s4;
                const float v = ldg(T.ew4 + (size_t)min(k, N - 1) * T.i1 + c);
                bw4[4 * g + s4] = (k < N) ? v : 0.f;
            }
    }
    if (euler0) {
        // layers 0 and 1 of the velocity field in one launch: the obs part of layer 0 (C0 = obs W0 + b0) is loop
        // invariant over the Euler steps, only the rank-(act+1) update with (a_s, t_s) changes.
        issue_b();
        const int ad = T.i2, apw = T.i1, M = T.M;
        // layer 0 as MFMA: [16 x 16] (a_s | t_s | 0) times the 16 rows of W0 that start at the action block (rows past
        // the t row are zero padding of the arena), accumulated onto C0.  Wave w owns column tiles w, w+4, ...
        // C0 / W0 fragment loads do not depend on a_s: issue them before folding the head partials.
        constexpr int CT = NA;      // column tiles per wave: K / 64 <= NA (NA covers 16 * K / 4 float4 over 256 threads)
        f32x4 cacc[CT];
        float wf[CT][4];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int ct = min(wave + 4 * t, (K >> 4) - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) cacc[t][i] = ldg(T.A + (size_t)(row0 + 4 * q + i) * T.lda + 16 * ct + c);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) wf[t][s4] = ldg(T.ew + (size_t)(4 * q + s4) * K + 16 * ct + c);
        }
        for (int e = tid; e < 16 * 16; e += FQL_THREADS) {
            const int r = e >> 4, j = e & 15;
            float a = 0.f;
            if (j < ad) {
                a = ldg(T.ea_in + (size_t)(row0 + r) * T.i0 + j);
                if (T.evp) {  // a_s = a_{s-1} + (sum of head partials + head bias) / flow_steps, fixed summation order
                    float pv[32];
#pragma unroll
                    for (int tp = 0; tp < 32; ++tp) pv[tp] = ldg(T.evp + ((size_t)min(tp, T.e_ntp - 1) * M + row0 + r) * apw + j);
                    float sum = 0.f;
#pragma unroll
                    for (int tp = 0; tp < 32; ++tp) sum += (tp < T.e_ntp) ? pv[tp] : 0.f;
                    a += (sum + ldg(T.eb + j)) * T.f0;
                }
                if (tn == 0 && T.ea_out) stg(T.ea_out + (size_t)(row0 + r) * apw + j, a);
            } else if (j == ad) {
                a = T.f1;  // t_s
            }
            ea[r * 32 + j] = a;
        }
        __syncthreads();
        {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&ea[c * 32 + 4 * q]);  // A'[row c][k = 4 q + s]
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                if (wave + 4 * t < (K >> 4)) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) cacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], wf[t][s4], cacc[t], 0, 0, 0);
                    const int ct = wave + 4 * t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) lds[(4 * q + i) * S + 16 * ct + c] = gelu_f(cacc[t][i]);
                }
            }
        }
    } else if (EUL && TMT == 1 && (flags & GF_A_LOSSACT)) {
        // head dgrad of the one-step actor: dA[r][a] = alpha 2/(B act) (a_raw - target) + [ -1 < a_raw < 1 ] (dQ/da member 0 + 1)
        // (agents/fql.py:66 distill, :69-72 clip mask on the Q path).  ea_in = a_raw, evp = target (ld i0), ew / ew4 = the
        // critic members' input gradients (ld lda of THEIR pass = e_ntp, action block at column i1), f0 = alpha 2 / (B act).
        issue_b();
        for (int e = tid; e < 16 * K; e += FQL_THREADS) {
            const int r = sdiv(e, frcp(K)), j = e - r * K;
            float g = 0.f;
            if (j < T.i2) {
                const float ar = ldg(T.ea_in + (size_t)(row0 + r) * T.i0 + j);
                float tg;
                if (flags & GF_A_EULFIN) {   // same loads, same summation order as fql_euler_finish_kernel: the target is bit-identical
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
                if (tn == 0 && T.ea_out) stg(T.ea_out + (size_t)(row0 + r) * T.i0 + j, g);   // the head's wgrad reads it
            }
            lds[r * S + j] = g;
        }
    } else {
#pragma unroll
    for (int pass = 0; pass < TMT; ++pass) {
        f32x4 av[NA];
        const float* __restrict__ Ag = T.A + (size_t)(row0 + 16 * pass) * T.lda;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = min(tid + i * FQL_THREADS, nA - 1);  // clamped, unconditional
            const int r = sdiv(f, rk4), kk = f - r * k4;
            av[i] = ldg4(Ag + (size_t)r * T.lda + 4 * kk);
        }
        if (pass == 0) issue_b();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = tid + i * FQL_THREADS;
            if (f < nA) {
                const int r = sdiv(f, rk4), kk = f - r * k4;
                *reinterpret_cast<f32x4*>(&lds[(16 * pass + r) * S + 4 * kk]) = av[i];
            }
        }
    }
    }
    STAMP();
    __syncthreads();
    STAMP();
    if (flags & GF_A_LN) {
        // 16 threads per row; flax LayerNorm: eps 1e-6, var = max(0, E[x^2] - E[x]^2)
        const int j = tid & 15;
        const int width = T.ln_width;
        const bool wr = (flags & GF_LN_WRITE) && tn == 0;
#pragma unroll
        for (int pass = 0; pass < TMT; ++pass) {
            const int r = 16 * pass + (tid >> 4);
            float s = 0.f, s2 = 0.f;
            // batches of 8 independent reads / loads: a rolled loop would pay one LDS (or, below, L2) round trip per element
            for (int kb = j; kb < width; kb += 128) {
                float xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = lds[r * S + min(kb + 16 * i, width - 1)];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (kb + 16 * i < width) { s += xv[i]; s2 += xv[i] * xv[i]; }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s += __shfl_xor(s, o);
                s2 += __shfl_xor(s2, o);
            }
            const float inv = 1.0f / (float)width;
            const float mean = s * inv;
            const float var = fmaxf(0.0f, s2 * inv - mean * mean);
            const float rstd = 1.0f / sqrtf(var + 1e-6f);
            for (int kb = j; kb < K; kb += 128) {
                float xv[8], gv[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kc = min(kb + 16 * i, width - 1);
                    gv[i] = ldg(T.ln_g + kc); bv[i] = ldg(T.ln_b + kc);
                    xv[i] = lds[r * S + kc];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = kb + 16 * i;
                    if (k >= K) continue;
                    const float v = (k < width) ? (xv[i] - mean) * rstd * gv[i] + bv[i] : 0.f;
                    lds[r * S + k] = v;
                    if (wr) stg(T.ln_xout + (size_t)(row0 + r) * T.lda + k, v);
                }
            }
            if (wr && j == 0) {
                stg(T.ln_stats + 2 * (row0 + r), mean);
                stg(T.ln_stats + 2 * (row0 + r) + 1, rstd);
            }
        }
        __syncthreads();
    }

    // epilogue operand of the dgrad forms (GELU' / ReLU mask of the stored pre-activation): fetched before the MFMA loop so its
    // L2 round trip is not the last thing on the launch's critical path
    float zpre[TMT][4];
#pragma unroll
    for (int rt = 0; rt < TMT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) zpre[rt][i] = 0.f;
    if (active && kp == 0 && (flags & (GF_GELUGRAD | GF_RELUGRAD))) {
#pragma unroll
        for (int rt = 0; rt < TMT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) zpre[rt][i] = ldg(T.Zprev + (size_t)(row0 + 16 * rt + 4 * q + i) * T.ldc + n0 + c);
    }
    f32x4 acc[TMT];
#pragma unroll
    for (int r = 0; r < TMT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        if (transb) gemm_wave<true, TMT>(acc, ba, lds, S, gbeg, gend, c, q, b0, b1);
        else gemm_wave<false, TMT>(acc, ba, lds, S, gbeg, gend, c, q, b0, b1);
    }
    STAMP();
    if (WK > 1) {
        if (kp > 0) {
#pragma unroll
            for (int r = 0; r < TMT; ++r) *reinterpret_cast<f32x4*>(&red[((((kp - 1) * NW + nt) * TMT + r) * 64 + lane) * 4]) = acc[r];
        }
        __syncthreads();
        if (kp > 0 && !head) return;
        if (kp == 0)
        for (int p = 1; p < WK; ++p) {
#pragma unroll
            for (int r = 0; r < TMT; ++r) acc[r] += *reinterpret_cast<const f32x4*>(&red[((((p - 1) * NW + nt) * TMT + r) * 64 + lane) * 4]);
        }
    }
    if (head) {
        // last hidden layer + action head in one launch: this workgroup's 16 x 32 GELU tile is multiplied into its 32
        // rows of the head kernel; the 16 x ap partial goes to evp[tn] and is folded by the next step's first launch.
        if (kp == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) hs[(4 * q + i) * 36 + 16 * nt + c] = active ? gelu_f(acc[0][i] + bias) : 0.f;
        }
        __syncthreads();
        if (wave == 0) {
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&hs[c * 36 + 16 * g + 4 * q]);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], bw4[4 * g + s4], pa, 0, 0, 0);
            }
            if (c < T.i1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) stg(T.evp + ((size_t)tn * T.M + row0 + 4 * q + i) * T.i1 + c, pa[i]);
            }
        }
        return;
    }
    if (!active) return;
    STAMP();

    // ---- epilogue. C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int n = n0 + c;
    if (flags & GF_C_FRAGT) {   // this lane holds rows 4 q + i of column n: four 4-byte stores (written once per update, read ten times)
#pragma unroll
        for (int rt = 0; rt < TMT; ++rt) {
            const int m0 = row0 + 16 * rt;
            float* tb = T.C + ((size_t)(m0 >> 4) * (T.ldc >> 4) + (n >> 4)) * 256 + ((n & 15) >> 2) * 64 + (n & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) stg(tb + (4 * q + i) * 4, acc[rt][i] + bias);
        }
        return;
    }
    if (flags & GF_C_FRAG) {
#pragma unroll
        for (int rt = 0; rt < TMT; ++rt) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = acc[rt][i] + bias;
            stg4(T.C + ((size_t)(((row0 + 16 * rt) >> 2) + q) * T.ldc + n) * 4, o);
        }
        return;
    }
#pragma unroll
    for (int rt = 0; rt < TMT; ++rt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 16 * rt + 4 * q + i;
        float v = acc[rt][i] + bias;
        const size_t o = (size_t)row * T.ldc + n;
        if (flags & GF_EULER) {
            // agents/fql.py:166-169: actions = actions + vels / flow_steps ; t = (i+1)/flow_steps
            float* xr = T.aux + (size_t)row * T.i0 + T.i1;
            if (n < T.i2) {
                const float a = ldg(xr + n) + v * T.f0;
                stg(xr + n, a);
                if (flags & GF_EULER_LAST) stg(T.aux2 + (size_t)row * T.ldc + n, clip1(a));
            }
            if (n == 0) stg(xr + T.i2, T.f1);
            continue;
        }
        if (flags & GF_SAVE_Z) { float gg, dg; gelu_both(v, gg, dg); stg(T.Zout + o, dg); v = (flags & GF_GELU) ? gg : v; }
        else if (flags & GF_GELU) v = gelu_f(v);
        if (flags & GF_GELUGRAD) v *= zpre[rt][i];
        if (flags & GF_RELUGRAD) v = (zpre[rt][i] > 0.f) ? v : 0.f;
        if (flags & GF_CLIP_OUT) v = clip1(v);
        stg(T.C + o, v);
        if ((flags & GF_OS_SCATTER) && n < T.i2) {
            // agents/fql.py:26 next_actions = clip(onestep(next_obs)) -> target-critic input; :69 clip(actor_actions) -> critic input
            const int B3 = T.M / 3;
            if (row < B3) stg(T.aux + (size_t)row * T.i0 + T.i1 + n, clip1(v));
            else if (row < 2 * B3) stg(T.aux2 + (size_t)(row - B3) * T.i0 + T.i1 + n, clip1(v));
        }
    }
    }
#ifdef FQL_STAMPS
    STAMP();
    if (lane == 0 && wave == 0 && T.aux) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(T.aux) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}

// TMT2: the launch contains tasks with two row tiles per workgroup; KBIG: some task has K > 512.  Separate
// kernels so the common case (K <= 512) keeps its register count (and with it two workgroups per CU).
#ifndef FQL_GEMM_WAVES
#define FQL_GEMM_WAVES 1
#endif
// (Passing the task by value in the kernel-argument segment was tried for single-task launches: slower -- the
// kernarg segment is host-visible memory and a 200-byte struct costs more than one hop through the HBM table.)
template <bool TMT2, bool KBIG>
__global__ __launch_bounds__(FQL_THREADS, FQL_GEMM_WAVES) void fql_gemm16_kernel(const GemmTask* __restrict__ tasks, int ntasks, int tl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    tl_enter(tl);
    const GemmTask& T = tasks[find_task(tasks, ntasks, blockIdx.x)];
    if (TMT2 && T.tmt == 2) {
        if (T.K <= 128) gemm16_body<2, 2, false>(T, lds);
        else if (!KBIG || T.K <= 512) gemm16_body<8, 2, false>(T, lds);
        else gemm16_body<16, 2, false>(T, lds);
    } else {
        if (T.K <= 128) gemm16_body<2, 1, false>(T, lds);
        else if (!KBIG || T.K <= 512) gemm16_body<8, 1, false>(T, lds);
        else gemm16_body<16, 1, false>(T, lds);
    }
    tl_exit(tl);
}
// Euler-chain launches (fused layer-0 build / head partial): own kernel so their registers do not tax the others
template <bool KBIG>
__global__ __launch_bounds__(FQL_THREADS) void fql_gemm16_euler_kernel(const GemmTask* __restrict__ tasks, int ntasks, int tl) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    tl_enter(tl);
    const GemmTask& T = tasks[find_task(tasks, ntasks, blockIdx.x)];
    if (T.K <= 128) gemm16_body<2, 1, true>(T, lds);
    else if (!KBIG || T.K <= 512) gemm16_body<8, 1, true>(T, lds);
    else gemm16_body<16, 1, true>(T, lds);
    tl_exit(tl);
}

// ------------------------------------------------------------------------------------------------
// Throughput lane GEMM: 64 x 64 output tile per workgroup, K streamed through LDS in 64-deep chunks
// (double buffered: chunk i+1 travels global -> VGPR while chunk i feeds the matrix cores), each wave a
// 32 x 32 sub-tile = 2 x 2 MFMA tiles with independent accumulators.  0.125 B of L2 traffic per MAC (the
// 16-row latency kernel needs 0.37) and 64 MFMAs per wave per chunk behind one barrier.
// LayerNorm on the A operand needs whole-row statistics while A arrives in K chunks, so the PRODUCING layer
// emits per-row partial sums (sum, sum of squares) per 64-column tile (GF_LN_PART) and the consumer folds
// them in fixed order (deterministic) into mean / rstd before staging (utils/networks.py:58).
// ------------------------------------------------------------------------------------------------
#define G64_S 68  // LDS row stride (floats): 64 + 4, keeps 16-byte alignment and spreads banks
// Fold the per-32-column LayerNorm partial sums (sum, sum of squares) of NR rows (row_first + 16 i) in fixed order: 2 partials per
// float4, batches of 4 float4 per row (two batches at width 512: the batch size bounds the live registers), rows padded to 16 bytes.
template <int NR>
__device__ __forceinline__ void ln_fold_partials(const float* __restrict__ partials, int row_first, int ntin, float (&sum)[NR], float (&sumsq)[NR]) {
    const int rs = (2 * ntin + 3) & ~3;
    const int nf4 = (ntin + 1) >> 1;
#pragma unroll
    for (int i = 0; i < NR; ++i) { sum[i] = 0.f; sumsq[i] = 0.f; }
    for (int base = 0; base < nf4; base += 4) {   // uniform trip count
        f32x4 pv[NR][4];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float* pp = partials + (size_t)(row_first + 16 * i) * rs;
#pragma unroll
            for (int t = 0; t < 4; ++t) pv[i][t] = ldg4(pp + 4 * min(base + t, nf4 - 1));
        }
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (2 * (base + t) < ntin) { sum[i] += pv[i][t][0]; sumsq[i] += pv[i][t][1]; }
                if (2 * (base + t) + 1 < ntin) { sum[i] += pv[i][t][2]; sumsq[i] += pv[i][t][3]; }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Throughput lane GEMM, 64 x 64 tile (batches >= 1024); see the header comment above
// ------------------------------------------------------------------------------------------------
template <bool transb, int RI>
__device__ __forceinline__ void gemm64_body(const GemmTask& T, float* lds) {
    constexpr int TM = 32 * RI;
    float* As = lds;                    // [2][TM][G64_S]
    float* Bs = lds + 2 * TM * G64_S;   // [2][64][G64_S]   ([k][n], or [n][k] when GF_TRANS_B)
    float* part = Bs + 2 * 64 * G64_S;  // [2][TM][2] LN partial sums of the epilogue
    const int local = (int)blockIdx.x - T.tile0;  // gemm64 tasks always come first in a launch
    const int tm = sdiv(local, frcp(T.ntn)), tn = local - tm * T.ntn;
    const int row0 = tm * TM, n0 = tn * 64;
    const int K = T.K;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int c = lane & 15, q = lane >> 4;
    const int flags = T.flags;
    const bool a_ln = (flags & GF_A_LN) != 0;
    const bool ln_wr = (flags & GF_LN_WRITE) && tn == 0;
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define GSTAMP() do { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP() do {} while (0)
#endif
    GSTAMP();
    // staging coordinates: f = tid + 256 i -> row f / 16, float4 column f % 16 (2 RI float4 per thread for A, 4 for B)
    const int sr = tid >> 4, sc4 = tid & 15;
    constexpr int NRA = 2 * RI;
    // two register sets: a chunk's loads stay in flight across two compute phases.  The loads carry NOTHING that waits for them (LayerNorm is applied
    // when a chunk is written to LDS, as in the 32-row body), and the first two chunks are requested before the statistics' partial sums are folded.
    f32x4 ra0[NRA], rb0[4], ra1[NRA], rb1[4], lg0, lb0, lg1, lb1;
    lg0 = lb0 = lg1 = lb1 = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ Ag = T.A + (size_t)(row0 + sr) * T.lda + 4 * sc4;
    const float* __restrict__ Bg = transb ? T.B + (size_t)(n0 + sr) * T.ldb + 4 * sc4 : T.B + (size_t)sr * T.ldb + n0 + 4 * sc4;
    auto load_chunk = [&](f32x4 (&ra)[NRA], f32x4 (&rb)[4], f32x4& lg, f32x4& lb, int k0) {  // K is a multiple of 64: no guards, unconditional loads
#pragma unroll
        for (int i = 0; i < NRA; ++i) ra[i] = ldg4(Ag + (size_t)(16 * i) * T.lda + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = transb ? ldg4(Bg + (size_t)(16 * i) * T.ldb + k0) : ldg4(Bg + (size_t)(k0 + 16 * i) * T.ldb);
        if (a_ln) { lg = ldg4(T.ln_g + k0 + 4 * sc4); lb = ldg4(T.ln_b + k0 + 4 * sc4); }
    };
    const int nchunks = K >> 6;
    load_chunk(ra0, rb0, lg0, lb0, 0);
    if (nchunks > 1) load_chunk(ra1, rb1, lg1, lb1, 64);
    float mean[NRA], rstd[NRA];
#pragma unroll
    for (int i = 0; i < NRA; ++i) { mean[i] = 0.f; rstd[i] = 1.f; }
    if (a_ln) {
        const float inv = 1.0f / (float)T.ln_width;
        float sm[NRA], sq[NRA];
        ln_fold_partials<NRA>(T.aux2, row0 + sr, T.i0, sm, sq);   // T.i0 = K / 32 partials per row
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int row = row0 + sr + 16 * i;
            mean[i] = sm[i] * inv;
            const float var = fmaxf(0.0f, sq[i] * inv - mean[i] * mean[i]);
            rstd[i] = 1.0f / sqrtf(var + 1e-6f);
            if (ln_wr && sc4 == 0) { stg(T.ln_stats + 2 * row, mean[i]); stg(T.ln_stats + 2 * row + 1, rstd[i]); }
        }
    }
    GSTAMP();   // [1] LayerNorm statistics folded
    auto store_chunk = [&](f32x4 (&ra)[NRA], const f32x4 (&rb)[4], const f32x4& lg, const f32x4& lb, int k0, int buf) {
        float* a = As + buf * TM * G64_S;
        float* b = Bs + buf * 64 * G64_S;
        if (a_ln) {   // utils/networks.py:58 on the rows of this chunk
            const int k = k0 + 4 * sc4;
#pragma unroll
            for (int i = 0; i < NRA; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (ra[i][e] - mean[i]) * rstd[i] * lg[e] + lb[e];
                    ra[i][e] = (k + e < T.ln_width) ? v : 0.f;
                }
                if (ln_wr) stg4(T.ln_xout + (size_t)(row0 + sr + 16 * i) * T.lda + k, ra[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NRA; ++i) *reinterpret_cast<f32x4*>(a + (sr + 16 * i) * G64_S + 4 * sc4) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(b + (sr + 16 * i) * G64_S + 4 * sc4) = rb[i];
    };
    f32x4 acc[RI][2];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float bias0 = (flags & GF_BIAS) ? ldg(T.bias + n0 + 32 * wc + c) : 0.f;
    const float bias1 = (flags & GF_BIAS) ? ldg(T.bias + n0 + 32 * wc + 16 + c) : 0.f;

    auto compute = [&](int buf) {
        const float* a = As + buf * TM * G64_S + (16 * RI * wr + c) * G64_S + 4 * q;
        const float* b = Bs + buf * 64 * G64_S;
        f32x4 fa[2][RI], fb[2][2];  // [pipeline slot][tile]: group g+1's fragments are read while group g multiplies
        auto read_frags = [&](int slot, int g) {
#pragma unroll
            for (int i = 0; i < RI; ++i) fa[slot][i] = *reinterpret_cast<const f32x4*>(a + 16 * i * G64_S + 16 * g);
            if (transb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[slot][j] = *reinterpret_cast<const f32x4*>(b + (32 * wc + 16 * j + c) * G64_S + 16 * g + 4 * q);
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) fb[slot][j][s] = b[(16 * g + 4 * q + s) * G64_S + 32 * wc + 16 * j + c];
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) read_frags((g + 1) & 1, g + 1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g & 1][i][s], fb[g & 1][j][s], acc[i][j], 0, 0, 0);
        }
    };
    // chunk ch computes from LDS slot ch & 1 while chunk ch+1 (set 0 / 1 alternating) and ch+2 are in flight
    store_chunk(ra0, rb0, lg0, lb0, 0, 0);
    if (nchunks > 2) load_chunk(ra0, rb0, lg0, lb0, 128);
    __syncthreads();
    GSTAMP();   // [2] first chunk staged (the stamp's wait also drains the two prefetched chunks: diagnostics only)
#ifdef FQL_STAMPS  // diagnostic ablations (never in the product build): bit 15 = no streaming loads, bit 16 = no MFMAs
    const bool dbg_noload = (flags >> 15) & 1, dbg_nocompute = (flags >> 16) & 1;
#else
    const bool dbg_noload = false, dbg_nocompute = false;
#endif
    for (int ch = 0; ch < nchunks; ch += 2) {
        // (register set 1 holds chunk ch + 1, set 0 chunk ch + 2 on entry)
        if (!dbg_nocompute) compute(0);
        if (ch + 1 < nchunks) store_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 1), 1);
        __syncthreads();
        if (ch + 3 < nchunks && !dbg_noload) load_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 3));
        if (ch + 1 < nchunks) {
            if (!dbg_nocompute) compute(1);
            if (ch + 2 < nchunks) store_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 2), 0);
            __syncthreads();
            if (ch + 4 < nchunks && !dbg_noload) load_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 4));
        }
    }

    GSTAMP();   // [3] K loop done
    // ---- epilogue. C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    float s1[RI][4], s2[RI][4];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * RI * wr + 16 * i + 4 * q + r;
            s1[i][r] = 0.f; s2[i][r] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + 32 * wc + 16 * j + c;
                const size_t o = (size_t)row * T.ldc + n;
                float v = acc[i][j][r] + (j ? bias1 : bias0);
                if (flags & GF_SAVE_Z) { float gg, dg; gelu_both(v, gg, dg); stg(T.Zout + o, dg); v = (flags & GF_GELU) ? gg : v; }
                else if (flags & GF_GELU) v = gelu_f(v);
                if (flags & GF_GELUGRAD) v *= ldg(T.Zprev + o);
                if (flags & GF_RELUGRAD) v = (ldg(T.Zprev + o) > 0.f) ? v : 0.f;
                stg(T.C + o, v);
                s1[i][r] += v; s2[i][r] += v * v;
            }
        }
    }
    if (flags & GF_LN_PART) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[i][r], b = s2[i][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (c == 0) {
                    const int rl = 16 * RI * wr + 16 * i + 4 * q + r;
                    part[(wc * TM + rl) * 2] = a;
                    part[(wc * TM + rl) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * TM) {   // one partial per 32 columns (the column half of a wave): T.i1 = N / 32 partials per row
            const int half = tid / TM, rl = tid - half * TM;
            float* pp = T.aux + (size_t)(row0 + rl) * ((2 * T.i1 + 3) & ~3) + 2 * (2 * tn + half);
            stg(pp, part[(half * TM + rl) * 2]);
            stg(pp + 1, part[(half * TM + rl) * 2 + 1]);
        }
    }
#ifdef FQL_STAMPS
    GSTAMP();   // [4] epilogue stored
    if (tid == 0 && T.stamps64) {
        unsigned long long* d = T.stamps64 + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Throughput-lane tile, round 2: 32 x 32 or 32 x 64 output tile (NJ = 1 | 2 MFMA column tiles per wave), K streamed through
// LDS in 64-deep chunks exactly as above.  Why a second shape: a level of the side lane is ~1536 rows x 512 columns = 384
// tiles of 32 x 64 on 256 CUs - half the CUs get two tiles, the level lasts two tile times; 768 tiles of 32 x 32 are three per
// CU on every CU (three co-resident workgroups also fill each other's barrier / LDS-store bubbles).  The engine picks the shape
// per launch from the tile count (schedule()).
// LayerNorm on the A operand is applied when a chunk is STORED to LDS, not when it is loaded: the loads of the next two chunks
// stay in flight across the MFMA phase (round 1 normalised at load time, which put a full L2 round trip in front of every
// chunk's MFMAs: 22.7 against 16.2 us per level), and the first chunks are requested before the row statistics are folded.
// LN partial sums are per 32 output columns whatever the tile shape, so producer and consumer need not agree on a shape.
// ------------------------------------------------------------------------------------------------
template <bool transb, int NJ>
__device__ __forceinline__ void gemm32_body(const GemmTask& T, float* lds) {
    constexpr int TN = 32 * NJ;
    constexpr int BS = transb ? G64_S : TN + 4;   // LDS row stride of a B chunk: [n][k] (dgrad) or [k][n]
    constexpr int BROWS = transb ? TN : 64;
    constexpr int NB = 2 * NJ;                    // float4 per thread of a B chunk
    float* As = lds;                              // [2][32][G64_S]
    float* Bs = lds + 2 * 32 * G64_S;             // [2][BROWS][BS]
    float* part = Bs + 2 * BROWS * BS;            // [2 column halves][32][2] LN partial sums of the epilogue
    const int local = (int)blockIdx.x - T.tile0;
    int tm = sdiv(local, frcp(T.ntn)), tn = local - tm * T.ntn;
    if (T.xg) xcd_tile(local, T.M >> 5, T.ntn, T.xg, tm, tn);
    const int row0 = tm * 32, n0 = tn * TN;
    const int K = T.K;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int c = lane & 15, q = lane >> 4;
    const int flags = T.flags;
    const bool a_ln = (flags & GF_A_LN) != 0;
    const bool ln_wr = (flags & GF_LN_WRITE) && tn == 0;
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define TSTAMP() do { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP() do {} while (0)
#endif
    TSTAMP();
    const int sr = tid >> 4, sc4 = tid & 15;      // A staging: rows sr, sr + 16; float4 column sc4
    // B staging coordinates of float4 i (row of the LDS image, float offset inside the row)
    const int br0 = (transb || NJ == 2) ? sr : (tid >> 3), brs = (transb || NJ == 2) ? 16 : 32;
    const int bc4 = (transb || NJ == 2) ? sc4 : (tid & 7);
    const float* __restrict__ Ag = T.A + (size_t)(row0 + sr) * T.lda + 4 * sc4;
    const float* __restrict__ Bg = transb ? T.B + (size_t)(n0 + br0) * T.ldb + 4 * bc4 : T.B + (size_t)br0 * T.ldb + n0 + 4 * bc4;
    f32x4 ra0[2], rb0[NB], ra1[2], rb1[NB], lg0, lb0, lg1, lb1;
    lg0 = lb0 = lg1 = lb1 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](f32x4 (&ra)[2], f32x4 (&rb)[NB], f32x4& lg, f32x4& lb, int k0) {   // K is a multiple of 64: no guards
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = ldg4(Ag + (size_t)(16 * i) * T.lda + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = transb ? ldg4(Bg + (size_t)(brs * i) * T.ldb + k0) : ldg4(Bg + (size_t)(k0 + brs * i) * T.ldb);
        if (a_ln) { lg = ldg4(T.ln_g + k0 + 4 * sc4); lb = ldg4(T.ln_b + k0 + 4 * sc4); }
    };
    const int nchunks = K >> 6;
    load_chunk(ra0, rb0, lg0, lb0, 0);
    if (nchunks > 1) load_chunk(ra1, rb1, lg1, lb1, 64);
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if (a_ln) {   // the partial sums travel behind the first two chunks; nothing waits for either before the fold below
        const float inv = 1.0f / (float)T.ln_width;
        float sm[2], sq[2];
        ln_fold_partials<2>(T.aux2, row0 + sr, T.i0, sm, sq);   // T.i0 = K / 32 partials per row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            mean[i] = sm[i] * inv;
            const float var = fmaxf(0.0f, sq[i] * inv - mean[i] * mean[i]);
            rstd[i] = 1.0f / sqrtf(var + 1e-6f);
            if (ln_wr && sc4 == 0) { const int row = row0 + sr + 16 * i; stg(T.ln_stats + 2 * row, mean[i]); stg(T.ln_stats + 2 * row + 1, rstd[i]); }
        }
    }
    TSTAMP();   // [1] LayerNorm statistics folded
    auto store_chunk = [&](f32x4 (&ra)[2], const f32x4 (&rb)[NB], const f32x4& lg, const f32x4& lb, int k0, int buf) {
        float* a = As + buf * 32 * G64_S;
        float* b = Bs + buf * BROWS * BS;
        if (a_ln) {   // utils/networks.py:58 on the rows of this chunk
            const int k = k0 + 4 * sc4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (ra[i][e] - mean[i]) * rstd[i] * lg[e] + lb[e];
                    ra[i][e] = (k + e < T.ln_width) ? v : 0.f;
                }
                if (ln_wr) stg4(T.ln_xout + (size_t)(row0 + sr + 16 * i) * T.lda + k, ra[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(a + (sr + 16 * i) * G64_S + 4 * sc4) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(b + (br0 + brs * i) * BS + 4 * bc4) = rb[i];
    };
    f32x4 acc[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The product runs TRANSPOSED (weight fragment as the A operand of the MFMA, activation fragment as B: the same registers, swapped), so lane (c, q)
    // holds output row 16 wr + c, columns 16 (wc NJ + j) + 4 q .. + 3: bias, GELU'(z) in / out and C are 16-byte accesses, and a row's LayerNorm
    // partial sums meet over the four q lanes (two shuffles).
    f32x4 bias[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bias[j] = (flags & GF_BIAS) ? ldg4(T.bias + n0 + (wc * NJ + j) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const float* a = As + buf * 32 * G64_S + (16 * wr + c) * G64_S + 4 * q;
        const float* b = Bs + buf * BROWS * BS;
        f32x4 fa[2], fb[2][NJ];   // [pipeline slot]: group g+1's fragments are read while group g multiplies
        auto read_frags = [&](int slot, int g) {
            fa[slot] = *reinterpret_cast<const f32x4*>(a + 16 * g);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = (wc * NJ + j) * 16 + c;
                if (transb) fb[slot][j] = *reinterpret_cast<const f32x4*>(b + col * BS + 16 * g + 4 * q);
                else {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) fb[slot][j][s4] = b[(16 * g + 4 * q + s4) * BS + col];
                }
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) read_frags((g + 1) & 1, g + 1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j][g & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[g & 1][j][s4], fa[g & 1][s4], acc[j][g & 1], 0, 0, 0);
        }
    };
    // chunk ch computes from LDS slot ch & 1 while chunks ch+1 (registers) and ch+2 (in flight) follow
    store_chunk(ra0, rb0, lg0, lb0, 0, 0);
    if (nchunks > 2) load_chunk(ra0, rb0, lg0, lb0, 128);
    __syncthreads();
    TSTAMP();   // [2] first chunk staged
    for (int ch = 0; ch < nchunks; ch += 2) {
        compute(0);
        if (ch + 1 < nchunks) store_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 1), 1);
        __syncthreads();
        if (ch + 3 < nchunks) load_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 3));
        if (ch + 1 < nchunks) {
            compute(1);
            if (ch + 2 < nchunks) store_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 2), 0);
            __syncthreads();
            if (ch + 4 < nchunks) load_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 4));
        }
    }
    TSTAMP();   // [3] K loop done

    // ---- epilogue. C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    float s1 = 0.f, s2 = 0.f;   // this lane's share of its row's (sum, sum of squares) over the wave's 16 NJ columns
    {
        const int row = row0 + 16 * wr + c;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wc * NJ + j) * 16 + 4 * q;
            const size_t o = (size_t)row * T.ldc + n;
            f32x4 v = acc[j][0] + acc[j][1] + bias[j];
            if (flags & GF_SAVE_Z) {
                f32x4 dgv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { float gg, dg; gelu_both(v[r], gg, dg); dgv[r] = dg; v[r] = (flags & GF_GELU) ? gg : v[r]; }
                stg4(T.Zout + o, dgv);
            } else if (flags & GF_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            }
            if (flags & GF_GELUGRAD) v *= ldg4(T.Zprev + o);
            if (flags & GF_RELUGRAD) {
                const f32x4 zp = ldg4(T.Zprev + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (zp[r] > 0.f) ? v[r] : 0.f;
            }
            stg4(T.C + o, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1 += v[r]; s2 += v[r] * v[r]; }
        }
    }
    if (flags & GF_LN_PART) {   // per-row (sum, sum of squares) of each 32-column half: T.i1 = N / 32 partials per row
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (q == 0) {
            const int rl = 16 * wr + c;
            part[(wc * 32 + rl) * 2] = s1;
            part[(wc * 32 + rl) * 2 + 1] = s2;
        }
        __syncthreads();
        const int rs = (2 * T.i1 + 3) & ~3;
        if (NJ == 2) {
            if (tid < 64) {
                const int half = tid >> 5, rl = tid & 31;
                float* pp = T.aux + (size_t)(row0 + rl) * rs + 2 * (2 * tn + half);
                stg(pp, part[(half * 32 + rl) * 2]);
                stg(pp + 1, part[(half * 32 + rl) * 2 + 1]);
            }
        } else if (tid < 32) {
            float* pp = T.aux + (size_t)(row0 + tid) * rs + 2 * tn;
            stg(pp, part[tid * 2] + part[(32 + tid) * 2]);
            stg(pp + 1, part[tid * 2 + 1] + part[(32 + tid) * 2 + 1]);
        }
    }
#ifdef FQL_STAMPS
    TSTAMP();   // [4] epilogue stored
    if (tid == 0 && T.stamps64) {
        unsigned long long* d = T.stamps64 + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}
#define FQL_TILE_LDS_FLOATS(NJ) (2 * 32 * G64_S + 2 * 64 * G64_S + 128)   /* upper bound over transb / NJ */

// ------------------------------------------------------------------------------------------------
// precision = 2: the same 32 x (32 NJ) tile with split-bf16 operands (see the top of this file).
// Global loads, the two-chunk register pipeline, LayerNorm-at-store and the epilogue are those of gemm32_body; what changes is
// what a chunk looks like in LDS and what the K loop issues:
//   * every fp32 value is split ONCE, by the thread that stages it, into a hi and a lo bf16 plane (same bytes as the fp32 chunk);
//   * A (and W^T for dgrads) planes are [row][64 k] with 32-word rows and no padding.  A lane's 16x16x32 fragment is 8 k values =
//     ONE ds_read_b128 per plane; the 4-word slots of a row are XOR-swizzled with (row >> 1) & 7, which makes both the b128
//     fragment reads (the hardware's 16-lane groups mix two k-quarters) and the b64 staging writes conflict-free
//     (tools/lds_banks.py enumerates the bank of every lane under the rules of MI355X_MICROARCH.md);
//   * the k order inside a 32-deep MFMA step is permuted - lane (c, q) owns k = 4q..4q+3 and 16+4q..16+4q+3 - identically for
//     both operands (a contraction does not care).  That makes the forward pass's B operand, stored [k][n] as it arrives from the
//     row-major kernel, readable with the hardware transpose ds_read_b64_tr_b16: a 32-lane half then covers 8 CONSECUTIVE k rows,
//     which the 8-word column chunks XOR-swizzled by the row spread over all 64 banks (no padding either);
//   * per 64-deep chunk a wave issues 2 x NJ x 3 MFMAs of 16 cycles (fp32 path: 16 NJ of 32 cycles); the hi x hi products
//     accumulate in acc[j][0], the two cross terms in acc[j][1], summed in the epilogue.
// ------------------------------------------------------------------------------------------------
template <bool transb, int NJ>
__device__ __forceinline__ void gemm32s_body(const GemmTask& T, float* lds_f) {
    constexpr int TN = 32 * NJ;
    constexpr int RW = TN / 2;                    // words per row of a [k][n] B plane
    constexpr int BPL = transb ? TN * 32 : 64 * RW;   // words per B plane
    constexpr int APL = 32 * 32;                  // words per A plane
    constexpr int NB = 2 * NJ;                    // float4 per thread of a B chunk
    unsigned* As = reinterpret_cast<unsigned*>(lds_f);   // [2 buffers][hi, lo][32 rows][32 words]
    unsigned* Bs = As + 4 * APL;                          // [2 buffers][hi, lo][BPL]
    float* part = reinterpret_cast<float*>(Bs + 4 * BPL); // [2 column halves][32][2] LN partial sums of the epilogue
    const int local = (int)blockIdx.x - T.tile0;
    int tm = sdiv(local, frcp(T.ntn)), tn = local - tm * T.ntn;
    if (T.xg) xcd_tile(local, T.M >> 5, T.ntn, T.xg, tm, tn);
    const int row0 = tm * 32, n0 = tn * TN;
    const int K = T.K;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int c = lane & 15, q = lane >> 4;
    const int flags = T.flags;
    const bool a_ln = (flags & GF_A_LN) != 0;
    const bool ln_wr = (flags & GF_LN_WRITE) && tn == 0;
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define SSTAMP() do { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SSTAMP() do {} while (0)
#endif
    SSTAMP();
    const int sr = tid >> 4, sc4 = tid & 15;      // A staging: rows sr, sr + 16; float4 column sc4 (k = 4 sc4 .. 4 sc4 + 3)
    const int br0 = (transb || NJ == 2) ? sr : (tid >> 3), brs = (transb || NJ == 2) ? 16 : 32;
    const int bc4 = (transb || NJ == 2) ? sc4 : (tid & 7);
    const float* __restrict__ Ag = T.A + (size_t)(row0 + sr) * T.lda + 4 * sc4;
    const float* __restrict__ Bg = transb ? T.B + (size_t)(n0 + br0) * T.ldb + 4 * bc4 : T.B + (size_t)br0 * T.ldb + n0 + 4 * bc4;
    f32x4 ra0[2], rb0[NB], ra1[2], rb1[NB], lg0, lb0, lg1, lb1;
    lg0 = lb0 = lg1 = lb1 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](f32x4 (&ra)[2], f32x4 (&rb)[NB], f32x4& lg, f32x4& lb, int k0) {   // K is a multiple of 64: no guards
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = ldg4(Ag + (size_t)(16 * i) * T.lda + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = transb ? ldg4(Bg + (size_t)(brs * i) * T.ldb + k0) : ldg4(Bg + (size_t)(k0 + brs * i) * T.ldb);
        if (a_ln) { lg = ldg4(T.ln_g + k0 + 4 * sc4); lb = ldg4(T.ln_b + k0 + 4 * sc4); }
    };
    const int nchunks = K >> 6;
    load_chunk(ra0, rb0, lg0, lb0, 0);
    if (nchunks > 1) load_chunk(ra1, rb1, lg1, lb1, 64);
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    if (a_ln) {
        const float inv = 1.0f / (float)T.ln_width;
        float sm[2], sq[2];
        ln_fold_partials<2>(T.aux2, row0 + sr, T.i0, sm, sq);   // T.i0 = K / 32 partials per row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            mean[i] = sm[i] * inv;
            const float var = fmaxf(0.0f, sq[i] * inv - mean[i] * mean[i]);
            rstd[i] = 1.0f / sqrtf(var + 1e-6f);
            if (ln_wr && sc4 == 0) { const int row = row0 + sr + 16 * i; stg(T.ln_stats + 2 * row, mean[i]); stg(T.ln_stats + 2 * row + 1, rstd[i]); }
        }
    }
    SSTAMP();   // [1] LayerNorm statistics folded
    // word offset, inside a 32-word [row][64 k] row, of the 4 k values a staging thread owns: 4-k block t = sc4 -> MFMA step
    // kp = t >> 3, owner quarter q = t & 3, half h = (t >> 2) & 1 of that lane's 8 values; slot 4 kp + q is swizzled per row
    const int st_slot = 4 * (sc4 >> 3) + (sc4 & 3), st_h = (sc4 >> 2) & 1;
    auto rowk_word = [&](int row) { return row * 32 + 4 * (st_slot ^ ((row >> 1) & 7)) + 2 * st_h; };
    auto store_chunk = [&](f32x4 (&ra)[2], const f32x4 (&rb)[NB], const f32x4& lg, const f32x4& lb, int k0, int buf) {
        unsigned* ah = As + buf * 2 * APL;
        unsigned* bh = Bs + buf * 2 * BPL;
        if (a_ln) {   // utils/networks.py:58 on the rows of this chunk
            const int k = k0 + 4 * sc4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (ra[i][e] - mean[i]) * rstd[i] * lg[e] + lb[e];
                    ra[i][e] = (k + e < T.ln_width) ? v : 0.f;
                }
                if (ln_wr) stg4(T.ln_xout + (size_t)(row0 + sr + 16 * i) * T.lda + k, ra[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2 hi, lo;
            bsplit4(ra[i], hi, lo);
            const int w = rowk_word(sr + 16 * i);
            *reinterpret_cast<u32x2*>(ah + w) = hi;
            *reinterpret_cast<u32x2*>(ah + APL + w) = lo;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            u32x2 hi, lo;
            bsplit4(rb[i], hi, lo);
            int w;
            if (transb) w = rowk_word(br0 + brs * i);
            else {   // [k][n] plane: row k = br0 + brs i, 4 columns n = 4 bc4 ..: 8-word chunk bc4 >> 2 swizzled by the row
                const int k = br0 + brs * i;
                const int key = (NJ == 2) ? ((k >> 1) & 3) : ((k >> 2) & 1);
                w = k * RW + 8 * ((bc4 >> 2) ^ key) + 2 * (bc4 & 3);
            }
            *reinterpret_cast<u32x2*>(bh + w) = hi;
            *reinterpret_cast<u32x2*>(bh + BPL + w) = lo;
        }
    };
    f32x4 acc[NJ][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The product runs TRANSPOSED (weight fragment as the A operand of the MFMA, activation fragment as B: the same registers, swapped), so lane (c, q)
    // holds output row 16 wr + c, columns 16 (wc NJ + j) + 4 q .. + 3: bias, GELU'(z) in / out and C are 16-byte accesses, and a row's LayerNorm
    // partial sums meet over the four q lanes (two shuffles).
    f32x4 bias[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bias[j] = (flags & GF_BIAS) ? ldg4(T.bias + n0 + (wc * NJ + j) * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses (words) that do not depend on the chunk
    const int a_row = (16 * wr + c) * 32, sw = (c >> 1) & 7;
    // transposed read: this lane supplies the address of row 4 q + qq (+ 16 h + 32 kp), columns 4 p .. 4 p + 3 of the wave's column tile
    const int qq = c >> 2, p = c & 3;
    const int tr_row = 4 * q + qq;
    const int tr_key = (NJ == 2) ? ((tr_row >> 1) & 3) : ((tr_row >> 2) & 1);
    auto compute = [&](int buf) {
        const unsigned* ah = As + buf * 2 * APL;
        const unsigned* bh = Bs + buf * 2 * BPL;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            const int slot = 4 * ((4 * kp + q) ^ sw);
            const u32x4 fah = *reinterpret_cast<const u32x4*>(ah + a_row + slot);
            const u32x4 fal = *reinterpret_cast<const u32x4*>(ah + APL + a_row + slot);
            u32x4 fbh[NJ], fbl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (transb) {
                    const int brow = ((wc * NJ + j) * 16 + c) * 32;   // (row >> 1) & 7 == (c >> 1) & 7: the tile base is a multiple of 16
                    fbh[j] = *reinterpret_cast<const u32x4*>(bh + brow + slot);
                    fbl[j] = *reinterpret_cast<const u32x4*>(bh + BPL + brow + slot);
                } else {
                    const int w0 = (32 * kp + tr_row) * RW + 8 * ((wc * NJ + j) ^ tr_key) + 2 * p;
                    const u32x2 h0 = lds_read_tr16(bh + w0), h1 = lds_read_tr16(bh + w0 + 16 * RW);
                    const u32x2 l0 = lds_read_tr16(bh + BPL + w0), l1 = lds_read_tr16(bh + BPL + w0 + 16 * RW);
                    fbh[j] = u32x4{h0[0], h0[1], h1[0], h1[1]};
                    fbl[j] = u32x4{l0[0], l0[1], l1[0], l1[1]};
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][0] = mfma_bf16(fbh[j], fah, acc[j][0]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][1] = mfma_bf16(fbl[j], fah, acc[j][1]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][1] = mfma_bf16(fbh[j], fal, acc[j][1]);
        }
    };
    // chunk ch computes from LDS slot ch & 1 while chunks ch+1 (registers) and ch+2 (in flight) follow
    store_chunk(ra0, rb0, lg0, lb0, 0, 0);
    if (nchunks > 2) load_chunk(ra0, rb0, lg0, lb0, 128);
    __syncthreads();
    SSTAMP();   // [2] first chunk staged
    for (int ch = 0; ch < nchunks; ch += 2) {
        compute(0);
        if (ch + 1 < nchunks) store_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 1), 1);
        __syncthreads();
        if (ch + 3 < nchunks) load_chunk(ra1, rb1, lg1, lb1, 64 * (ch + 3));
        if (ch + 1 < nchunks) {
            compute(1);
            if (ch + 2 < nchunks) store_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 2), 0);
            __syncthreads();
            if (ch + 4 < nchunks) load_chunk(ra0, rb0, lg0, lb0, 64 * (ch + 4));
        }
    }

    SSTAMP();   // [3] K loop done
    // ---- epilogue (that of gemm32_body). C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    float s1 = 0.f, s2 = 0.f;   // this lane's share of its row's (sum, sum of squares) over the wave's 16 NJ columns
    {
        const int row = row0 + 16 * wr + c;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wc * NJ + j) * 16 + 4 * q;
            const size_t o = (size_t)row * T.ldc + n;
            f32x4 v = acc[j][0] + acc[j][1] + bias[j];
            if (flags & GF_SAVE_Z) {
                f32x4 dgv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { float gg, dg; gelu_both(v[r], gg, dg); dgv[r] = dg; v[r] = (flags & GF_GELU) ? gg : v[r]; }
                stg4(T.Zout + o, dgv);
            } else if (flags & GF_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            }
            if (flags & GF_GELUGRAD) v *= ldg4(T.Zprev + o);
            if (flags & GF_RELUGRAD) {
                const f32x4 zp = ldg4(T.Zprev + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (zp[r] > 0.f) ? v[r] : 0.f;
            }
            stg4(T.C + o, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1 += v[r]; s2 += v[r] * v[r]; }
        }
    }
    if (flags & GF_LN_PART) {   // per-row (sum, sum of squares) of each 32-column half: T.i1 = N / 32 partials per row
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (q == 0) {
            const int rl = 16 * wr + c;
            part[(wc * 32 + rl) * 2] = s1;
            part[(wc * 32 + rl) * 2 + 1] = s2;
        }
        __syncthreads();
        const int rs = (2 * T.i1 + 3) & ~3;
        if (NJ == 2) {
            if (tid < 64) {
                const int half = tid >> 5, rl = tid & 31;
                float* pp = T.aux + (size_t)(row0 + rl) * rs + 2 * (2 * tn + half);
                stg(pp, part[(half * 32 + rl) * 2]);
                stg(pp + 1, part[(half * 32 + rl) * 2 + 1]);
            }
        } else if (tid < 32) {
            float* pp = T.aux + (size_t)(row0 + tid) * rs + 2 * tn;
            stg(pp, part[tid * 2] + part[(32 + tid) * 2]);
            stg(pp + 1, part[tid * 2 + 1] + part[(32 + tid) * 2 + 1]);
        }
    }
#ifdef FQL_STAMPS
    SSTAMP();   // [4] epilogue stored
    if (tid == 0 && T.stamps64) {
        unsigned long long* d = T.stamps64 + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}
#define FQL_TILE_SPLIT_LDS_FLOATS(NJ) (4 * 32 * 32 + 4 * 32 * 32 * (NJ) + 128)

// one throughput-lane tile task: T.tmt = 2 -> 64 x 64 tile (round-1 body; BIG launches only: it needs twice the registers, and
// the 32-row shapes want three or four workgroups per CU), else 32 x (32 T.wk)
template <bool BIG, bool SPLIT = false>
__device__ __forceinline__ void gemm_tile_dispatch(const GemmTask& T, float* lds) {
    if (SPLIT) {
        if (T.wk == 2) {
            if (T.flags & GF_TRANS_B) gemm32s_body<true, 2>(T, lds);
            else gemm32s_body<false, 2>(T, lds);
        } else {
            if (T.flags & GF_TRANS_B) gemm32s_body<true, 1>(T, lds);
            else gemm32s_body<false, 1>(T, lds);
        }
        return;
    }
    if (BIG && T.tmt == 2) {
        if (T.flags & GF_TRANS_B) gemm64_body<true, 2>(T, lds);
        else gemm64_body<false, 2>(T, lds);
    } else if (T.wk == 2) {
        if (T.flags & GF_TRANS_B) gemm32_body<true, 2>(T, lds);
        else gemm32_body<false, 2>(T, lds);
    } else {
        if (T.flags & GF_TRANS_B) gemm32_body<true, 1>(T, lds);
        else gemm32_body<false, 1>(T, lds);
    }
}

__global__ __launch_bounds__(FQL_THREADS, 2) void fql_gemm64_kernel(const GemmTask* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ti = find_task(tasks, ntasks, blockIdx.x);
    gemm_tile_dispatch<true>(tasks[ti], lds);
}

// ------------------------------------------------------------------------------------------------
// K9 wgrad: dW[Kin, N] = X^T dZ (contraction over the batch), db[n] = sum_m dZ[m, n]
//   implied by jax.grad, utils/flax_utils.py:137.
// Workgroup tile 32 (Kin) x 64 (N).  The 4 waves split the batch (contraction) dimension, each holds 2 x 4 16x16 accumulators.
// The product is transposed (dZ is the A operand, X the B operand) and BOTH operands' free indices are PERMUTED so that a lane
// loads whole vectors: lane (c, q) loads the 16 bytes dZ[m][n0 + 4 c .. + 3] - the 16 lanes of a row group read one whole
// 256-byte tile row, full 128-byte lines instead of four 64-byte segments per dword load - and feeds component t to the
// accumulators [.][t] (row i of accumulator t is column n0 + 4 i + t); it loads the 8 bytes X[m][k0 + 2 c .. + 1] and feeds
// component u to the accumulators [u][.] (column j of accumulator u is input k0 + 2 j + u).  Two loads feed eight MFMAs.
// A lane then owns dW[k0 + 2 c + u][n0 + 16 q + 4 r + t] (r = accumulator register): sixteen consecutive columns of two rows.
// Loads are issued a chunk of 8 steps (48 VGPRs) ahead; partial tiles meet in LDS and thread (wave w, lane) finalises register
// r = w of the eight accumulators = two 16-byte stores.  Summation order is fixed: gradients are bitwise reproducible.
// ------------------------------------------------------------------------------------------------
#define FQL_WGRAD_KT 32   // inputs (rows of dW) per workgroup tile; the engine counts tiles with it
#define FQL_WGRAD_LDS_FLOATS (4 * 8 * 64 * 4 + 4 * 64)
// (no per-lane validity here: a lane whose columns / inputs lie outside the matrix reads a clamped, valid address and computes values that only
// reach accumulator rows / columns the finish never stores)
__device__ __forceinline__ void wgrad_load_chunk(f32x2 (&a)[8], f32x4 (&b)[8], const float* xp, const float* zp, unsigned xo, unsigned zo, size_t sx, size_t sz, int cnt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool v = i < cnt;   // wave-uniform
        a[i] = v ? *(const FQL_GAS f32x2*)(xp + i * sx + xo) : f32x2{0.f, 0.f};
        b[i] = v ? ldg4(zp + i * sz + zo) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
// the reduction over the 4 waves and the stores, shared by the fp32 and the split body
template <int NU>   // NU inputs per lane: the tile is 16 NU inputs high
__device__ __forceinline__ void wgrad_finish(const WgradTask& T, float* lds, const f32x4 (&acc)[NU][4], const float (&bs)[4], int tk, int k0, int n0,
                                             int ntv, bool xv) {
    float* red = lds;                    // [wave][accumulator u, t][lane] float4
    float (*redb)[64] = reinterpret_cast<float (*)[64]>(lds + 4 * 4 * NU * 64 * 4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < NU; ++u) *reinterpret_cast<f32x4*>(&red[((wave * 4 * NU + 4 * u + t) * 64 + lane) * 4]) = acc[u][t];
        float v = bs[t];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (q == 0) redb[wave][4 * c + t] = v;
    }
    __syncthreads();
    if (xv && 4 * q + wave < 4 * ntv) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x4 r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = red[((0 * 4 * NU + 4 * u + t) * 64 + lane) * 4 + wave];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += red[((w * 4 * NU + 4 * u + t) * 64 + lane) * 4 + wave];
                r[t] = v;
            }
            stg4(T.dW + (size_t)(k0 + NU * c + u) * T.ldw + n0 + 16 * q + 4 * wave, r);
        }
    }
    if (tk == 0 && T.db && wave == 0 && lane < 16 * ntv) T.db[n0 + lane] = redb[0][lane] + redb[1][lane] + redb[2][lane] + redb[3][lane];
}
// lds: FQL_WGRAD_LDS_FLOATS floats.  bid = block index inside the wgrad task space of the launch.
__device__ __forceinline__ void wgrad_body(const WgradTask& T, int bid, float* lds) {
    const int local = bid - T.tile0;
    const int tk = sdiv(local, frcp(T.ntn)), tn = local - tk * T.ntn;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave-uniform row offsets stay in SGPRs
    const int c = lane & 15, q = lane >> 4;
    const int k0 = tk * FQL_WGRAD_KT, n0 = tn * 64;
    const int ntv = min(4, (T.N - n0) >> 4);  // valid 16-column tiles in this workgroup
    const bool xv = k0 + 2 * c < T.Kin;       // this lane's two inputs exist (Kin is a multiple of 16)
    const int steps = T.M >> 4;               // MFMA steps (4 batch rows each) per wave
    const size_t sx = (size_t)4 * T.ldx, sz = (size_t)4 * T.ldz;
    // uniform base (SGPR pair) + 32-bit lane offset: every load of a chunk shares one offset register
    const float* __restrict__ xp = T.X + (size_t)(4 * wave * steps) * T.ldx;    // B[k = m][j], j <-> inputs k0 + 2 j + u
    const float* __restrict__ zp = T.dZ + (size_t)(4 * wave * steps) * T.ldz;   // A[i][k = m], i <-> columns n0 + 4 i + t
    const unsigned xo = (unsigned)(q * T.ldx + min(k0 + 2 * c, T.Kin - 2)), zo = (unsigned)(q * T.ldz + n0 + min(4 * c, 16 * ntv - 4));
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[0][t] = acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2 a0[8], a1[8];
    f32x4 b0[8], b1[8];
    auto mma = [&](const f32x2(&a)[8], const f32x4(&b)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bs[t] += b[i][t];
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][t], a[i][0], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[i][t], a[i][1], acc[1][t], 0, 0, 0);
            }
    };
    wgrad_load_chunk(a0, b0, xp, zp, xo, zo, sx, sz, steps);
    for (int s = 0; s < steps; s += 16) {
        if (s + 8 < steps) wgrad_load_chunk(a1, b1, xp + (s + 8) * sx, zp + (s + 8) * sz, xo, zo, sx, sz, steps - s - 8);
        mma(a0, b0);
        if (s + 8 < steps) {
            if (s + 16 < steps) wgrad_load_chunk(a0, b0, xp + (s + 16) * sx, zp + (s + 16) * sz, xo, zo, sx, sz, steps - s - 16);
            mma(a1, b1);
        }
    }
    wgrad_finish<2>(T, lds, acc, bs, tk, k0, n0, ntv, xv);
}
// precision = 2 weight gradient: 16 x 64 tiles (FQL_WGRAD_KT_SPLIT: one input per lane - with two, as above, the bf16x3 update measured
// 2.5 % slower although the serialised launches were not), same dZ permutation, LDS reduction and summation order over waves.  The 8
// batch rows a lane holds per chunk for X^T and for each dZ column are exactly one 16x16x32 operand each (the k order inside an MFMA
// step is free as long as both operands share it), so a chunk is split in registers and multiplied with 3 x 4 bf16 MFMAs instead of
// 32 fp32 ones; db stays an fp32 column sum.  Loads are branch-free (out-of-range steps / columns re-read a valid address and are zeroed).
#define FQL_WGRAD_KT_SPLIT 16
__device__ __forceinline__ void wgrad_split_body(const WgradTask& T, int bid, float* lds) {
    const int local = bid - T.tile0;
    const int tk = sdiv(local, frcp(T.ntn)), tn = local - tk * T.ntn;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    const int k0 = tk * FQL_WGRAD_KT_SPLIT, n0 = tn * 64;
    const int ntv = min(4, (T.N - n0) >> 4);  // valid 16-column tiles in this workgroup
    const bool zv = c < 4 * ntv;
    const int steps = T.M >> 4;               // MFMA steps (4 batch rows each) per wave
    const size_t sx = (size_t)4 * T.ldx, sz = (size_t)4 * T.ldz;
    const float* __restrict__ xp = T.X + (size_t)(4 * wave * steps) * T.ldx;    // uniform bases + 32-bit lane offsets, as in wgrad_body
    const float* __restrict__ zp = T.dZ + (size_t)(4 * wave * steps) * T.ldz;
    const unsigned xo = (unsigned)(q * T.ldx + k0 + c), zo = (unsigned)(q * T.ldz + n0 + min(4 * c, 16 * ntv - 4));
    f32x4 acc[1][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float a0[8], a1[8];
    f32x4 b0[8], b1[8];
    auto load = [&](float (&a)[8], f32x4 (&b)[8], int s0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int st = min(s0 + i, steps - 1);
            a[i] = ldg(xp + st * sx + xo);
            b[i] = ldg4(zp + st * sz + zo);
        }
    };
    auto mma = [&](const float (&a)[8], const f32x4 (&b)[8], int s0) {
        float av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = (s0 + i < steps) ? a[i] : 0.f;   // zero X rows: the products of a clamped step vanish
        u32x4 ah, al;
#pragma unroll
        for (int i = 0; i < 4; ++i) { unsigned h, l; bsplit2(av[2 * i], av[2 * i + 1], h, l); ah[i] = h; al[i] = l; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 bh, bl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned h, l;
                const float b0v = (s0 + 2 * i < steps && zv) ? b[2 * i][t] : 0.f;
                const float b1v = (s0 + 2 * i + 1 < steps && zv) ? b[2 * i + 1][t] : 0.f;
                bs[t] += b0v + b1v;
                bsplit2(b0v, b1v, h, l);
                bh[i] = h; bl[i] = l;
            }
            acc[0][t] = mfma_bf16(bh, al, acc[0][t]);
            acc[0][t] = mfma_bf16(bl, ah, acc[0][t]);
            acc[0][t] = mfma_bf16(bh, ah, acc[0][t]);
            __builtin_amdgcn_sched_barrier(0);   // keeps the compiler from hoisting every split in front of the MFMAs (spills)
        }
    };
    load(a0, b0, 0);
    for (int s = 0; s < steps; s += 16) {
        if (s + 8 < steps) load(a1, b1, s + 8);
        mma(a0, b0, s);
        if (s + 8 < steps) {
            if (s + 16 < steps) load(a0, b0, s + 16);
            mma(a1, b1, s + 8);
        }
    }
    wgrad_finish<1>(T, lds, acc, bs, tk, k0, n0, ntv, true);
}
__global__ __launch_bounds__(FQL_THREADS) void fql_wgrad_kernel(const WgradTask* __restrict__ tasks, int ntasks) {
    __shared__ __attribute__((aligned(16))) float lds_w[FQL_WGRAD_LDS_FLOATS];
    wgrad_body(tasks[find_task(tasks, ntasks, blockIdx.x)], blockIdx.x, lds_w);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward fused with GELU' (critic, utils/networks.py:56-58 reversed):
//   dxhat = dY*gamma ; dg = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) ; dZ = dg GELU'(z)
// row tiles: one wave per row, the row lives in registers (one tanh per element).  Column tiles
// (param grads): dgamma = sum_m dY xhat, dbeta = sum_m dY; one workgroup per 16 columns, 16 row
// groups per workgroup, fixed summation order (deterministic).
// ------------------------------------------------------------------------------------------------
template <bool SYN>  // SYN: dY[m][k] = dq[m] * wq[k] (scalar head), else dY is read from memory
__device__ __forceinline__ void lnbwd_body(const LnBwdTask& T, int bid, float (*red)[16][16]) {
    const int local = bid - T.tile0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (local < T.ntiles_rows) {
        const int row = local * 4 + wave;
        if (row >= T.M) return;
        const float mean = ldg(T.stats + 2 * row), rstd = ldg(T.stats + 2 * row + 1);
        const float* dy = T.dY + (size_t)row * T.ld;
        const float* z = T.Z + (size_t)row * T.ld;
        const float* gv = T.Gv + (size_t)row * T.ld;
        const float dqr = SYN ? ldg(T.dq + (size_t)row * T.ldq) : 0.f;
        // H <= 1024 (a multiple of 16): <= 4 float4 per lane and tensor, columns 4 (lane + 64 i) ..; every load is issued first.  (Round 1 read 16
        // single floats per lane and tensor whatever H was - at H = 512 half of those 64 loads were clamped duplicates.)
        f32x4 zz[4], dd[4], gm[4], gq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k4 = 4 * (lane + 64 * i);
            if (k4 < T.H) {
                zz[i] = ldg4(z + k4);
                gq[i] = ldg4(gv + k4);
                gm[i] = ldg4(T.gamma + k4);
                if (SYN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dd[i][e] = dqr * ldg(T.wq + (size_t)(k4 + e) * T.ldw);
                } else dd[i] = ldg4(dy + k4);
            } else zz[i] = gq[i] = gm[i] = dd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 d[4], xh[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * (lane + 64 * i) + e;
                xh[i][e] = (gq[i][e] - mean) * rstd;
                d[i][e] = dd[i][e] * gm[i][e];
                if (k < T.width) { s1 += d[i][e]; s2 += d[i][e] * xh[i][e]; }
            }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        const float inv = 1.0f / (float)T.width;
        const float m1 = s1 * inv, m2 = s2 * inv;
        float* dz = T.dZ + (size_t)row * T.ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k4 = 4 * (lane + 64 * i);
            if (k4 < T.H) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (k4 + e < T.width) ? rstd * (d[i][e] - m1 - xh[i][e] * m2) * zz[i][e] : 0.f;
                stg4(dz + k4, o);
            }
        }
    } else {
        const int cc = threadIdx.x & 15, rg = threadIdx.x >> 4;
        const int col = (local - T.ntiles_rows) * 16 + cc;
        float sg = 0.f, sb = 0.f;
        if (col < T.width) {
            const float wc = SYN ? ldg(T.wq + (size_t)col * T.ldw) : 0.f;
            for (int m0 = rg; m0 < T.M; m0 += 256) {  // 16 rows per thread in flight: one round trip at M = 256 (was four rounds of 4 rows)
                f32x2 st[16];
                float dv[16], zv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int m = min(m0 + 16 * u, T.M - 1);
                    st[u] = *(const FQL_GAS f32x2*)(T.stats + 2 * m);
                    dv[u] = SYN ? ldg(T.dq + (size_t)m * T.ldq) * wc : ldg(T.dY + (size_t)m * T.ld + col);
                    zv[u] = ldg(T.Gv + (size_t)m * T.ld + col);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (m0 + 16 * u < T.M) {
                        const float xh = (zv[u] - st[u][0]) * st[u][1];
                        sg += dv[u] * xh; sb += dv[u];
                    }
                }
            }
        }
        red[0][rg][cc] = sg; red[1][rg][cc] = sb;
        __syncthreads();
        if (rg == 0 && col < T.H) {
            float g = 0.f, b = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { g += red[0][r][cc]; b += red[1][r][cc]; }
            stg(T.dgamma + col, (col < T.width) ? g : 0.f);
            stg(T.dbeta + col, (col < T.width) ? b : 0.f);
        }
    }
}
__global__ __launch_bounds__(FQL_THREADS) void fql_lnbwd_kernel(const LnBwdTask* __restrict__ tasks, int ntasks) {
    __shared__ float red[2][16][16];
    const int ti = find_task(tasks, ntasks, blockIdx.x);
    const LnBwdTask& T = tasks[ti];
    if (T.dq) lnbwd_body<true>(T, blockIdx.x, red);
    else lnbwd_body<false>(T, blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------------
// step state + batch assembly
// ------------------------------------------------------------------------------------------------
struct DevState {
    uint64_t rng_step;   // advanced once per update (keys the Philox streams)
    uint64_t rng_stream; // mixed into the Philox key: 0 by default, the rank in data-parallel runs (fql_set_rng_stream)
    int64_t adam_count;  // optax count
    int64_t train_step;  // TrainState.step
    double b1pow, b2pow; // 0.9^count, 0.999^count
    float grad_scale;    // 1/world for data-parallel mean
    float lam;           // normalize_q_loss factor of the current step
    float info[16];      // the 13 info scalars (+ scratch)
    float gmax_bits_pad; // unused
    int gmax, gmin;      // ordered-int encodings for atomicMax/Min
    float leaf_sumsq[256];
};

struct SrcDesc {  // where the batch comes from; rewritten by the host only when it changes
    const float *obs, *act, *rew, *mask, *nobs;
    const int64_t* idx;  // gather indices or null (rows 0..B-1 of the arrays above)
    const float *eps1, *x0, *t, *z, *eps2;  // null => engine RNG
    int64_t lo, span;    // RNG index range [lo, lo+span) when idx == null and use_rng_idx
    int use_rng_idx;
    int advance;         // 1: this launch advances rng_step/adam bookkeeping (update), 0: loss-only
    // balanced sampling (main.py:255-259): batch rows [split, B) come from a second set of arrays (the replay ring); 0 = one source.
    // idx (when given) holds indices into the first source for rows < split and into the second for the rest.
    int split;
    const float *obs2, *act2, *rew2, *mask2, *nobs2;
    int64_t lo2, span2;
};

struct PrepArgs {
    const SrcDesc* src;
    DevState* st;
    uint64_t key;
    int B, od, ad;
    int inp_c, inp_b;  // padded input widths: critic/onestep (od+ad), bc_flow (od+ad+1)
    int ap;            // padded action width (ld of the [B, ap] action-shaped buffers)
    float *X_os, *X_bc, *X_eu, *X_c1, *X_c2, *X_ct, *vel, *w_rew, *w_mask, *w_act;
    float* X_e0;  // [B, inp_b] observations only (fused Euler chain: loop-invariant part of layer 0) or null
    // visual agents: the "observation" block of each network input is that module's encoding of the batch images
    // (agents/fql.py:196-202; [B, od] each, row b = batch row b; E_os holds [obs ; next_obs] = 2B rows); null otherwise
    const float *E_c, *E_t, *E_bc, *E_os;
    int tl;   // timeline id (diagnostics)
    int part; // 0: every output; 1: the critical lane's inputs only (X_eu, X_e0); 2: everything else (one launch per lane: neither lane
              //    then waits for the other at the start of the update)
    unsigned* xsync;  // the XCD-resident Euler chain behind this launch (fql_xchain.h): its arrival flags and tickets (32-word slots 0..15; slot 16 is the
                      // sticky error word) are zeroed here, or null
};

// agents/fql.py:52-56 (x_t, vel), :144-150 (noise), utils/datasets.py:64-100 (index draw + gather),
// utils/networks.py:191,229-231 (concatenate) -- one pass builds every network input of the step.
__global__ __launch_bounds__(FQL_THREADS) void fql_prep_kernel(PrepArgs P) {
    tl_enter(P.tl);
    if (P.xsync && blockIdx.x == 0 && P.part != 2)
        for (int i = threadIdx.x; i < 16 * 32; i += FQL_THREADS) P.xsync[i] = 0u;
    const SrcDesc& S = *P.src;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= P.B) return;
    const uint64_t step = P.st->rng_step;
    const uint64_t key = P.key ^ (P.st->rng_stream * 0x9E3779B97F4A7C15ull);   // same create seed on every rank, different draws
    int64_t src = b;
    const bool second = S.split > 0 && b >= S.split;   // the replay half of a balanced batch
    if (S.idx) src = S.idx[b];
    else if (S.use_rng_idx) {
        const uint64_t u = rng_u32(key, step, 7u, (uint32_t)b);
        src = second ? S.lo2 + (int64_t)((u * (uint64_t)S.span2) >> 32) : S.lo + (int64_t)((u * (uint64_t)S.span) >> 32);
    }
    const int od = P.od, ad = P.ad, B = P.B;
    const bool vis = P.E_c != nullptr;
    const float* obs = vis ? nullptr : (second ? S.obs2 : S.obs) + (size_t)src * od;
    const float* nobs = vis ? nullptr : (second ? S.nobs2 : S.nobs) + (size_t)src * od;
    const float* act = (second ? S.act2 : S.act) + (size_t)src * ad;
    const float tt = S.t ? S.t[b] : rng_uniform(key, step, 3u, (uint32_t)b);
    const int maxw = P.inp_c > P.inp_b ? P.inp_c : P.inp_b;
    for (int j = lane; j < maxw; j += 64) {
        const bool is_obs = j < od, is_act = (j >= od) && (j < od + ad);
        const int a = j - od;
        float o = 0.f, no = 0.f, av = 0.f, e1 = 0.f, e2 = 0.f, zz = 0.f, xx = 0.f;
        float o_c, o_bc, o_os, no_os, no_t;  // per consumer: critic(obs), bc_flow(obs), onestep(obs), onestep(next), target(next)
        if (is_obs && !vis) { o = obs[j]; no = nobs[j]; }
        o_c = o_bc = o_os = o; no_os = no_t = no;
        if (is_obs && vis) {
            const size_t r = (size_t)b * od + j;
            o_c = P.E_c[r]; o_bc = P.E_bc[r]; no_t = P.E_t[r];
            o_os = P.E_os[r]; no_os = P.E_os[(size_t)B * od + r];
        }
        if (is_act) {
            av = act[a];
            e1 = S.eps1 ? S.eps1[(size_t)b * ad + a] : rng_normal(key, step, 1u, (uint32_t)b, (uint32_t)a);
            xx = S.x0 ? S.x0[(size_t)b * ad + a] : rng_normal(key, step, 2u, (uint32_t)b, (uint32_t)a);
            zz = S.z ? S.z[(size_t)b * ad + a] : rng_normal(key, step, 4u, (uint32_t)b, (uint32_t)a);
            e2 = S.eps2 ? S.eps2[(size_t)b * ad + a] : rng_normal(key, step, 5u, (uint32_t)b, (uint32_t)a);
        }
        if (j < P.inp_c && P.part != 1) {
            const size_t w = P.inp_c;
            P.X_os[(size_t)b * w + j] = is_obs ? no_os : e1;         // sample_actions(next_obs)  fql.py:25
            P.X_os[(size_t)(B + b) * w + j] = is_obs ? o_os : zz;    // onestep(obs, noises)      fql.py:65
            P.X_os[(size_t)(2 * B + b) * w + j] = is_obs ? o_os : e2;  // sample_actions(obs)     fql.py:82
            P.X_c1[(size_t)b * w + j] = is_obs ? o_c : av;           // critic(obs, actions)      fql.py:36
            P.X_c2[(size_t)b * w + j] = is_obs ? o_c : 0.f;          // action block filled after onestep
            P.X_ct[(size_t)b * w + j] = is_obs ? no_t : 0.f;
        }
        if (j < P.inp_b) {
            const size_t w = P.inp_b;
            const float xt = (1.0f - tt) * xx + tt * av;             // fql.py:55
            if (P.part != 1) P.X_bc[(size_t)b * w + j] = is_obs ? o_bc : (is_act ? xt : (j == od + ad ? tt : 0.f));
            if (P.part != 2) {
                P.X_eu[(size_t)b * w + j] = is_obs ? o_bc : (is_act ? zz : 0.f);  // t_0 = 0; encoded once (fql.py:162-163)
                if (P.X_e0) P.X_e0[(size_t)b * w + j] = is_obs ? o_bc : 0.f;
            }
        }
        if (is_act && P.part != 1) {
            P.vel[(size_t)b * P.ap + a] = av - xx;                     // fql.py:56
            P.w_act[(size_t)b * P.ap + a] = av;
        }
    }
    if (lane == 0 && P.part != 1) {
        P.w_rew[b] = (second ? S.rew2 : S.rew)[src];
        P.w_mask[b] = (second ? S.mask2 : S.mask)[src];
    }
}

// ------------------------------------------------------------------------------------------------
// single-workgroup loss / bookkeeping kernels (B <= a few thousand rows: latency, not bandwidth)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

struct PostOsArgs {
    const float* A_os;  // [3B, 16] raw one-step actor outputs
    const float* w_act; // [B, 16]
    float *X_ct, *X_c2; // [B, inp_c]
    DevState* st;
    int B, od, ad, inp_c, ap;
};
// agents/fql.py:26 (clip next actions), :69 (clip actor actions), :82-83 (mse metric)
__device__ __forceinline__ void fql_post_onestep_body(const PostOsArgs& P, float* sh) {
    float se = 0.f;
    const int n = P.B * P.ad;
    for (int e = threadIdx.x; e < n; e += FQL_THREADS) {
        const int b = sdiv(e, frcp(P.ad)), a = e - b * P.ad;
        const float d = clip1(P.A_os[(size_t)(2 * P.B + b) * P.ap + a]) - P.w_act[(size_t)b * P.ap + a];
        se += d * d;
    }
    const float tot = block_sum(se, sh);
    if (threadIdx.x == 0) P.st->info[9] = tot / (float)n;
}

struct LossCriticArgs {
    const float *q1a, *q1b, *tqa, *tqb;  // [B,16] column 0
    const float *w_rew, *w_mask;
    float *dq1a, *dq1b;                  // [B,16] column 0 (other columns stay zero)
    DevState* st;
    int B, q_agg, want_grad;
    float discount;
};
// agents/fql.py:28-44
__device__ __forceinline__ void fql_loss_critic_body(const LossCriticArgs& P, float* sh) {
    float sl = 0.f, sq = 0.f, mx = -INFINITY, mn = INFINITY;
    const float gs = 1.0f / (float)P.B;  // d/dq of mean over 2B of (q-y)^2 = 2 (q-y) / (2B)
    for (int b = threadIdx.x; b < P.B; b += FQL_THREADS) {
        const float ta = P.tqa[(size_t)b * 16], tb = P.tqb[(size_t)b * 16];
        const float nq = P.q_agg ? fminf(ta, tb) : 0.5f * (ta + tb);
        const float y = P.w_rew[b] + P.discount * P.w_mask[b] * nq;
        const float qa = P.q1a[(size_t)b * 16], qb = P.q1b[(size_t)b * 16];
        const float da = qa - y, db = qb - y;
        sl += da * da + db * db;
        sq += qa + qb;
        mx = fmaxf(mx, fmaxf(qa, qb));
        mn = fminf(mn, fminf(qa, qb));
        if (P.want_grad) {
            P.dq1a[(size_t)b * 16] = da * gs;
            P.dq1b[(size_t)b * 16] = db * gs;
        }
    }
    const float tl = block_sum(sl, sh), tq = block_sum(sq, sh);
    const float tmx = block_max(mx, sh), tmn = -block_max(-mn, sh);
    if (threadIdx.x == 0) {
        const float inv = 1.0f / (2.0f * (float)P.B);
        P.st->info[0] = tl * inv;
        P.st->info[1] = tq * inv;
        P.st->info[2] = tmx;
        P.st->info[3] = tmn;
    }
}

struct LossQArgs {
    const float *q2a, *q2b;  // [B,16] col 0: critic(obs, clip(actor_actions))
    float *dq2a, *dq2b;
    DevState* st;
    int B, normalize, want_grad;
};
// agents/fql.py:70-76
__device__ __forceinline__ void fql_loss_q_body(const LossQArgs& P, float* sh) {
    float s = 0.f, sa = 0.f;
    for (int b = threadIdx.x; b < P.B; b += FQL_THREADS) {
        const float q = 0.5f * (P.q2a[(size_t)b * 16] + P.q2b[(size_t)b * 16]);
        s += q; sa += fabsf(q);
    }
    const float ts = block_sum(s, sh), tsa = block_sum(sa, sh);
    const float qmean = ts / (float)P.B;
    const float lam = P.normalize ? 1.0f / (tsa / (float)P.B) : 1.0f;
    if (P.want_grad) {
        const float g = -lam / (2.0f * (float)P.B);
        for (int b = threadIdx.x; b < P.B; b += FQL_THREADS) {
            P.dq2a[(size_t)b * 16] = g;
            P.dq2b[(size_t)b * 16] = g;
        }
    }
    if (threadIdx.x == 0) {
        P.st->info[7] = lam * (-qmean);
        P.st->info[8] = qmean;
        P.st->lam = lam;
    }
}

struct LossBcArgs {
    const float *pred, *vel;  // [B,16]
    float* dpred;             // [B,16]
    DevState* st;
    int B, ad, ap, want_grad;
};
// agents/fql.py:58-59
__device__ __forceinline__ void fql_loss_bc_body(const LossBcArgs& P, float* sh) {
    const int n = P.B * P.ad;
    const float gs = 2.0f / (float)n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += FQL_THREADS) {
        const int b = sdiv(e, frcp(P.ad)), a = e - b * P.ad;
        const float d = P.pred[(size_t)b * P.ap + a] - P.vel[(size_t)b * P.ap + a];
        s += d * d;
        if (P.want_grad) P.dpred[(size_t)b * P.ap + a] = gs * d;
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) P.st->info[5] = t / (float)n;
}

struct LossActorArgs {
    const float* a_raw;      // [B,16] one-step actor output rows (obs, z)
    const float* tgt;        // [B,16] clip(Euler(bc_flow))
    const float *dxa, *dxb;  // [B, inp_c] critic input gradients of the two members (or null)
    float* da;               // [B,16]
    DevState* st;
    int B, od, ad, inp_c, ap, want_grad;
    float alpha;
};
// agents/fql.py:66 (distill), :69-79 (clip mask on the Q path, total actor loss)
__device__ __forceinline__ void fql_loss_actor_body(const LossActorArgs& P, float* sh) {
    const int n = P.B * P.ad;
    const float gs = P.alpha * 2.0f / (float)n;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += FQL_THREADS) {
        const int b = sdiv(e, frcp(P.ad)), a = e - b * P.ad;
        const float ar = P.a_raw[(size_t)b * P.ap + a];
        const float d = ar - P.tgt[(size_t)b * P.ap + a];
        s += d * d;
        if (P.want_grad) {
            float g = gs * d;
            if (ar > -1.0f && ar < 1.0f) {
                const size_t o = (size_t)b * P.inp_c + P.od + a;
                g += P.dxa[o] + P.dxb[o];
            }
            P.da[(size_t)b * P.ap + a] = g;
        }
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const float distill = t / (float)n;
        P.st->info[6] = distill;
        P.st->info[4] = P.st->info[5] + P.alpha * distill + P.st->info[7];
    }
}

__global__ __launch_bounds__(FQL_THREADS) void fql_post_onestep_kernel(PostOsArgs P) { __shared__ float sh[4]; fql_post_onestep_body(P, sh); }
__global__ __launch_bounds__(FQL_THREADS) void fql_loss_critic_kernel(LossCriticArgs P) { __shared__ float sh[4]; fql_loss_critic_body(P, sh); }
__global__ __launch_bounds__(FQL_THREADS) void fql_loss_q_kernel(LossQArgs P) { __shared__ float sh[4]; fql_loss_q_body(P, sh); }
__global__ __launch_bounds__(FQL_THREADS) void fql_loss_bc_kernel(LossBcArgs P) { __shared__ float sh[4]; fql_loss_bc_body(P, sh); }
__global__ __launch_bounds__(FQL_THREADS) void fql_loss_actor_kernel(LossActorArgs P) { __shared__ float sh[4]; fql_loss_actor_body(P, sh); }

// single-workgroup loss / metric tasks that ride in a side-lane launch
enum : int { MISC_POSTOS = 0, MISC_LOSS_CRITIC, MISC_LOSS_Q, MISC_LOSS_BC };
struct MiscTask {
    int kind;
    union {
        PostOsArgs po;
        LossCriticArgs lc;
        LossQArgs lq;
        LossBcArgs lb;
    };
};

// ------------------------------------------------------------------------------------------------
// K10-K12: grad stats + Adam + Polyak in one pass over the trainable arena
//   utils/flax_utils.py:139-157 (stats), optax.adam (agents/fql.py:237), agents/fql.py:113-120 (Polyak
//   reads the PRE-step critic: p_old is still in a register when the target is written).
// ------------------------------------------------------------------------------------------------
struct AdamChunk { int off, len, leaf; };
struct AdamArgs {
    float *P, *G, *Mu, *Nu, *T;  // T = target arena (same layout as the critic block at offset 0)
    const AdamChunk* chunks;
    DevState* st;
    float* partials;  // [n_chunks][4]: sum of squares, max, min of the (scaled) gradient chunk
    int chunk0;       // first chunk of this launch (per-module launches of the fused single-GPU update)
    int critic_size;
    float lr, tau;
    int tl;
    const unsigned* err;   // persistent Euler chain only: its time-out word; set => the update's targets are invalid, leave the parameters alone
    int n0, chunk1;        // two modules in one launch: workgroups [0, n0) take chunks chunk0 + b, the rest chunk1 + (b - n0); n0 < 0: one range
};
__device__ __forceinline__ int f2ord(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(FQL_THREADS) void fql_adam_kernel(AdamArgs A) {
    __shared__ float sh[4];
    tl_enter(A.tl);
    if (A.err && __hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (null unless FQL_PEC=1)
    const int cidx = (A.n0 < 0 || (int)blockIdx.x < A.n0) ? (int)blockIdx.x + A.chunk0 : (int)blockIdx.x - A.n0 + A.chunk1;
    const AdamChunk ch = A.chunks[cidx];  // <= 4096 elements, offset and length multiples of 4
    // optax bias correction with count = adam_count + 1 (the counters advance in the finalize kernel afterwards)
    const float c1 = (float)(1.0 - A.st->b1pow * 0.9), c2 = (float)(1.0 - A.st->b2pow * 0.999);
    const float gsc = A.st->grad_scale;
    float ss = 0.f, mx = -INFINITY, mn = INFINITY;
    const int n4 = ch.len >> 2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = threadIdx.x + it * FQL_THREADS;
        if (i < n4) {
            const int o = ch.off + 4 * i;
            f32x4 g = ldg4(A.G + o);
            const f32x4 p = ldg4(A.P + o);
            f32x4 m = ldg4(A.Mu + o);
            f32x4 v = ldg4(A.Nu + o);
            f32x4 pn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] * gsc;
                m[e] = 0.9f * m[e] + 0.1f * ge;
                v[e] = 0.999f * v[e] + 0.001f * ge * ge;
                const float mh = m[e] / c1, vh = v[e] / c2;
                pn[e] = p[e] - A.lr * (mh / (sqrtf(vh) + 1e-8f));
                ss += ge * ge; mx = fmaxf(mx, ge); mn = fminf(mn, ge);
            }
            stg4(A.Mu + o, m);
            stg4(A.Nu + o, v);
            stg4(A.P + o, pn);
            if (o < A.critic_size) {
                f32x4 t = ldg4(A.T + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = p[e] * A.tau + t[e] * (1.0f - A.tau);
                stg4(A.T + o, t);
            }
        }
    }
    // per-chunk partials, folded in fixed order by the finalize kernel: no atomics, bitwise reproducible stats
    const float tss = block_sum(ss, sh);
    const float tmx = block_max(mx, sh), tmn = -block_max(-mn, sh);
    if (threadIdx.x == 0) {
        float* pp = A.partials + 4 * (size_t)cidx;
        pp[0] = tss; pp[1] = tmx; pp[2] = tmn;
    }
    tl_exit(A.tl);
}

struct FinalizeArgs {
    DevState* st;
    const AdamChunk* chunks;
    const float* partials;   // [n_chunks][4]
    const int* leaf_range;   // [nleaves + 1] chunk index ranges (chunks of a leaf are contiguous)
    int n_chunks, nleaves, do_grad_stats;
    int tl;
};
// grad/max, grad/min, grad/norm = sum over leaves of ||g_leaf||_2 (utils/flax_utils.py:139-157); target-critic leaves
// contribute zeros, which bound grad/max >= 0 >= grad/min (F5).  Also advances optax count / TrainState.step / RNG step.
__global__ __launch_bounds__(FQL_THREADS) void fql_finalize_kernel(FinalizeArgs A) {
    __shared__ float sh[4];
    __shared__ float leafn[256];
    DevState* st = A.st;
    tl_enter(A.tl);
    if (!A.do_grad_stats) return;
    float mx = -INFINITY, mn = INFINITY;
    for (int cidx = threadIdx.x; cidx < A.n_chunks; cidx += FQL_THREADS) {
        mx = fmaxf(mx, A.partials[4 * cidx + 1]);
        mn = fminf(mn, A.partials[4 * cidx + 2]);
    }
    // 4 threads per leaf, each a contiguous quarter of the leaf's chunks, combined in fixed order
    for (int base = 0; base < A.nleaves; base += 64) {
        const int l = base + (threadIdx.x >> 2), sub = threadIdx.x & 3;
        float s = 0.f;
        if (l < A.nleaves) {
            const int b = A.leaf_range[l], e = A.leaf_range[l + 1];
            const int per = (e - b + 3) >> 2;
            const int lo = b + sub * per, hi = min(e, lo + per);
            for (int cidx = lo; cidx < hi; ++cidx) s += A.partials[4 * cidx];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (l < A.nleaves && sub == 0) leafn[l] = sqrtf(s);
    }
    const float tmx = block_max(mx, sh), tmn = -block_max(-mn, sh);
    __syncthreads();
    if (threadIdx.x == 0) {
        float nrm = 0.f;
        for (int i = 0; i < A.nleaves; ++i) nrm += leafn[i];
        st->info[10] = fmaxf(tmx, 0.0f);
        st->info[11] = fminf(tmn, 0.0f);
        st->info[12] = nrm;
        st->rng_step += 1;
        st->adam_count += 1;
        st->train_step += 1;
        st->b1pow *= 0.9;
        st->b2pow *= 0.999;
    }
}

// fused Euler chain, last step: target = clip(a_{n-1} + (sum of head partials + bias) / flow_steps)  (agents/fql.py:169-170)
struct EulerFinishArgs {
    const float* a_in;   // [M, ap]
    const float* evp;    // [ntp][M][ap]
    const float* eb;     // [ap]
    float* tgt;          // [M, ap]
    int M, ad, ap, ntp, a_ld;
    float scale;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_euler_finish_kernel(EulerFinishArgs P) {
    const int e = blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e >= P.M * P.ad) return;
    const int r = sdiv(e, frcp(P.ad)), j = e - r * P.ad;
    float pv[32];
#pragma unroll
    for (int tp = 0; tp < 32; ++tp) pv[tp] = ldg(P.evp + ((size_t)min(tp, P.ntp - 1) * P.M + r) * P.ap + j);
    float sum = 0.f;
#pragma unroll
    for (int tp = 0; tp < 32; ++tp) sum += (tp < P.ntp) ? pv[tp] : 0.f;
    stg(P.tgt + (size_t)r * P.ap + j, clip1(ldg(P.a_in + (size_t)r * P.a_ld + j) + (sum + ldg(P.eb + j)) * P.scale));
}

// ------------------------------------------------------------------------------------------------
// Persistent Euler chain (agents/fql.py:155-171): the whole 10-step x (layers 0..3 + head) chain in ONE launch.
//
// A kernel boundary costs ~1.6 us plus ~2.5 us of cold first loads (the per-XCD L2s are written back and
// invalidated at every boundary), 30 times per update.  Here the H/32 workgroups that own one 16-row tile of the
// batch (a "team": one 32-column slice of every hidden layer each) hand their 16 x 32 output tiles to each other
// as 8-byte {tag, value} granules (CDNA4 guide, Guideline 16 form R2: the data is the flag):
//   producer: one naturally aligned 8-byte agent-scope relaxed atomic store per value, straight from the MFMA
//             accumulator layout -- no drain, no flag, no counter;
//   consumer: every thread sweeps its granules of the team's 16 x H tile with 8-byte agent-scope relaxed atomic
//             loads until every tag equals the phase tag (wall-clock bounded), then stages the values in LDS.
// Tags never repeat: tag = (launch epoch of the team << 6) + phase index + 1, the epoch living in device memory and
// advanced by member 0 when it has consumed the team's last phase (every member has read it by then), so neither
// a memset node nor a per-launch argument is needed and a graph replay is safe.  A buffer is rewritten only after
// every member has passed the phase that read it (the phase order implies it: a member can publish phase p + 2
// only after it has gathered phase p + 1 from all members, each of which gathered phase p before publishing).
// Results do not depend on placement; teams are mapped to one XCD only for speed.  A sweep that times out sets
// an error word and falls through, so the grid always drains.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long fql_u64;
struct PecArgs {
    const float* C0;      // [M, H]  obs W0 + b0 (loop invariant)
    const float* a0;      // [M, lda0] initial actions (noise z): X_eu + obs_dim
    const float* W0act;   // W0 rows of (action block, t): [ad + 1 (+ zero rows)][H]
    const float* W[3];    // hidden kernels of layers 1..3, [H][H]
    const float* b[3];    // their biases
    const float* W4;      // head kernel [H][ap]
    const float* b4;      // head bias [ap]
    fql_u64* G[2];        // [M][H] activation granules, ping-pong
    fql_u64* Vg;          // [T][M][16] head-partial granules
    float* tgt;           // [M, ap] out: clip(a_n)
    unsigned* epoch;      // [teams] launch epochs (device resident, advanced by the kernel)
    unsigned* err;        // set to 1 if a sweep timed out
    int M, ad, ap, lda0, flow_steps, nteams, ntile;  // nteams teams, each owning ntile 16-row tiles (team + nteams j)
};

__device__ __forceinline__ fql_u64 ld_granule(const fql_u64* p) {
    return __hip_atomic_load((const FQL_GAS fql_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(fql_u64* p, unsigned tag, float v) {
    __hip_atomic_store((FQL_GAS fql_u64*)p, ((fql_u64)tag << 32) | (fql_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform: true when a spin has lasted 200 ms of wall clock (100 MHz ticks): something is badly wrong, drain the grid
__device__ __forceinline__ bool pec_spin_fail(fql_u64& t0, unsigned* err) {
    const fql_u64 now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) { t0 = now; return false; }
    if (now - t0 > 20000000ull) {
        __hip_atomic_store((FQL_GAS unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

#define PEC_MAX_TILES 8
template <int H>  // hidden width (all four hidden layers), multiple of 256; T = H / 32 members per team
__global__ __launch_bounds__(FQL_THREADS) void fql_euler_persistent_kernel(const PecArgs P) {
    constexpr int T = H / 32, S = H + 4, G2 = H / 32;  // G2 = k-groups (of 16) per K-half
    constexpr int CT = H / 64;                          // layer-0 column tiles per wave
    constexpr int KPR = H / FQL_THREADS;                // granule sweeps per row of the activation tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* red = lds + 16 * S;   // [2 column tiles][64] float4: K-split partials
    float* hs = red + 1024;      // [16][36] staging: last-hidden tile for the head partial
    float* eas = hs + 576;       // [ntile][16][32] current actions a_s of every row tile (persistent across steps)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int nt = wave & 1, kp = wave >> 1;
    // team / member mapping: members of a team share blockIdx % 8 (one XCD under round-robin placement: speed only)
    int team, mem;
    if (P.nteams % 8 == 0) {
        const int x = blockIdx.x & 7, qq = blockIdx.x >> 3;
        mem = qq % T; team = x + 8 * (qq / T);
    } else {
        team = blockIdx.x / T; mem = blockIdx.x % T;
    }
    // a team owns the 16-row tiles team, team + nteams, ...: every phase runs over all of them in turn, so the
    // granules a tile needs were published a whole round earlier and their flight time hides behind the other tiles
    const int ntile = P.ntile;
    const int n0 = mem * 32 + nt * 16;
    const int M = P.M, ad = P.ad, ap = P.ap;
    const unsigned ep = __hip_atomic_load((FQL_GAS unsigned*)(P.epoch + team), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tag0 = ep << 6;  // phase tags of this launch: tag0 + 1 .. tag0 + 3 flow_steps (< 64)

    // ---- operands.  The hidden-layer B fragments (K-half x 16 columns = 4 G2 VGPRs) are fetched once per phase (for
    // all row tiles), BEFORE the sweep for the team's tile (they do not depend on it) and from an L2 that is never
    // invalidated inside this launch; keeping all three layers resident instead (192 VGPRs) would push the kernel
    // to 450 VGPRs and evict every other lane's waves from the CUs for the length of the chain.
    float wb[4 * G2];  // element (k = 16 (kp G2 + g) + 4 q + s, n0 + c) of the current layer
    auto load_w = [&](const float* Wl) {
        // the per-lane base is made opaque so the 4 G2 load addresses are formed here, next to the loads: hoisted
        // out of the step loop they would occupy >100 VGPR pairs for the whole kernel
        const float* base = Wl + (size_t)(16 * kp * G2 + 4 * q) * H + n0 + c;
        asm volatile("" : "+v"(base));
#pragma unroll
        for (int g = 0; g < G2; ++g)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) wb[4 * g + s4] = ldg(base + (size_t)(16 * g + s4) * H);
    };
    float bias[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) bias[l] = ldg(P.b[l] + n0 + c);
    float bw4[8];       // head kernel rows of this workgroup's 32 hidden columns (used by wave 0)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bw4[4 * g + s4] = ldg(P.W4 + (size_t)(mem * 32 + 16 * g + 4 * q + s4) * ap + c);
    // a_0 (+ zero fill of the [16][32] blocks); column ad carries t_s
    for (int e = tid; e < ntile * 512; e += FQL_THREADS) {
        const int j = e >> 9, r = (e >> 5) & 15, col = e & 31;
        eas[e] = (col < ad) ? ldg(P.a0 + (size_t)((team + P.nteams * j) * 16 + r) * P.lda0 + col) : 0.f;
    }
    const float inv_steps = 1.0f / (float)P.flow_steps;

    auto gemm_phase = [&](const float (&wl)[4 * G2]) -> f32x4 {  // A tile in LDS -> 16 x 16 accumulator (kp 0 holds the sum)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        const float* arow = lds + c * S + 4 * q + 16 * kp * G2;
#pragma unroll
        for (int g = 0; g < G2; g += 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * g);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(arow + 16 * g + 16);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], wl[4 * g + s4], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s4], wl[4 * g + 4 + s4], acc2, 0, 0, 0);
            }
        }
        acc += acc2;
        if (kp == 1) *reinterpret_cast<f32x4*>(&red[(nt * 64 + lane) * 4]) = acc;
        __syncthreads();
        if (kp == 0) acc += *reinterpret_cast<const f32x4*>(&red[(nt * 64 + lane) * 4]);
        return acc;
    };
    // GELU tile -> granules, straight from the accumulator layout (col = c, rows 4 q + i): 16 lanes = 128 contiguous bytes
    auto publish_tile = [&](const f32x4& acc, float bl, fql_u64* dst, int row0, unsigned tag) {
        if (kp == 0) {
            fql_u64* d = dst + (size_t)(row0 + 4 * q) * H + n0 + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) st_granule(d + (size_t)i * H, tag, gelu_f(acc[i] + bl));
        }
    };
    // a 16 x H tile: thread t sweeps granules t + 256 k (row k / KPR) until all carry `tag`, then stages them in LDS
    auto gather_tile = [&](const fql_u64* src, int row0, unsigned tag) {
#ifndef FQL_PEC_NO_SENTINEL
        // hint poll (guide: one idle wave polls, with s_sleep): lanes 0..T-1 of wave 0 watch the LAST granule each member
        // stores (row 15, last column of its slice) before anybody sweeps; the sweep below still checks every tag, so this
        // only keeps 64 KB sweeps of a tile that is not there yet off the CU's memory pipe while other kernels share it
        // (with >= 4 row tiles per team the tile was published a whole round ago and the hint only costs a barrier: skipped)
        if (ntile < 4 && wave == 0) {
            const fql_u64* sp = src + (size_t)(row0 + 15) * H + min(lane, T - 1) * 32 + 31;
            fql_u64 t1 = 0;
            for (;;) {
                const bool ok = lane >= T || (unsigned)(ld_granule(sp) >> 32) == tag;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if (pec_spin_fail(t1, P.err)) break;
            }
        }
        if (ntile < 4) __syncthreads();
#endif
        const fql_u64* sb = src + (size_t)row0 * H + tid;
        asm volatile("" : "+v"(sb));
        constexpr int NG = 16 * KPR;
        fql_u64 v[NG];
        fql_u64 t0 = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                v[i] = ld_granule(sb + (size_t)i * FQL_THREADS);
                ok &= (unsigned)(v[i] >> 32) == tag;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (pec_spin_fail(t0, P.err)) break;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) lds[(i / KPR) * S + tid + FQL_THREADS * (i % KPR)] = __uint_as_float((unsigned)v[i]);
    };
    // fold the T head partials of the previous phase C in fixed order: thread (r, j) owns action element (row0 + r, j)
    auto gather_head = [&](int row0, unsigned tag) -> float {
        const int r = tid >> 4, j = tid & 15;
        const bool act = j < ad;
        const fql_u64* vb = P.Vg + ((size_t)row0 + r) * 16 + (act ? j : 0);
        fql_u64 v[T];
        fql_u64 t0 = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                v[tp] = ld_granule(vb + (size_t)tp * M * 16);
                ok &= (unsigned)(v[tp] >> 32) == tag;
            }
            if (__all(ok || !act)) break;
            __builtin_amdgcn_s_sleep(1);
            if (pec_spin_fail(t0, P.err)) break;
        }
        float sum = 0.f;
#pragma unroll
        for (int tp = 0; tp < T; ++tp) sum += __uint_as_float((unsigned)v[tp]);
        return sum;
    };

#pragma clang loop unroll(disable)
    for (int s = 0; s < P.flow_steps; ++s) {
        const unsigned tagA = tag0 + 3 * s + 1, tagB = tagA + 1, tagC = tagA + 2;
        // ---------------- phase A: fold head partials -> a_s ; layer 0 (rank update of C0) ; layer 1
        load_w(P.W[0]);
        float wf[CT][4];    // W0 action/t rows for this wave's layer-0 column tiles (wave + 4 t)
        {
            const float* wb0 = P.W0act + (size_t)(4 * q) * H + 16 * wave + c;
            asm volatile("" : "+v"(wb0));
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) wf[t][s4] = ldg(wb0 + (size_t)s4 * H + 64 * t);
        }
#pragma clang loop unroll(disable)
        for (int j = 0; j < ntile; ++j) {
            const int row0 = (team + P.nteams * j) * 16;
            float* ea = eas + 512 * j;
            f32x4 c0f[CT];      // C0 in C layout for the same column tiles
            {
                const float* cb = P.C0 + (size_t)(row0 + 4 * q) * H + 16 * wave + c;
                asm volatile("" : "+v"(cb));
#pragma unroll
                for (int t = 0; t < CT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c0f[t][i] = ldg(cb + (size_t)i * H + 64 * t);
            }
            if (s > 0) {
                const float sum = gather_head(row0, tagA - 1);  // phase C of step s - 1
                const int r = tid >> 4, jj = tid & 15;
                if (jj < ad) ea[r * 32 + jj] += (sum + ldg(P.b4 + jj)) * inv_steps;
            }
            if (tid < 16) ea[tid * 32 + ad] = (float)s * inv_steps;  // t_s
            __syncthreads();
            {
                const f32x4 af = *reinterpret_cast<const f32x4*>(&ea[c * 32 + 4 * q]);
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    f32x4 h = c0f[t];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) h = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4], wf[t][s4], h, 0, 0, 0);
                    const int ct = wave + 4 * t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) lds[(4 * q + i) * S + 16 * ct + c] = gelu_f(h[i]);
                }
            }
            __syncthreads();
            const f32x4 acc = gemm_phase(wb);
            publish_tile(acc, bias[0], P.G[0], row0, tagA);
        }
        // ---------------- phase B: layer 2
        load_w(P.W[1]);
#pragma clang loop unroll(disable)
        for (int j = 0; j < ntile; ++j) {
            const int row0 = (team + P.nteams * j) * 16;
            gather_tile(P.G[0], row0, tagA);
            __syncthreads();
            const f32x4 acc = gemm_phase(wb);
            publish_tile(acc, bias[1], P.G[1], row0, tagB);
        }
        // ---------------- phase C: layer 3 + head partial
        load_w(P.W[2]);
#pragma clang loop unroll(disable)
        for (int j = 0; j < ntile; ++j) {
            const int row0 = (team + P.nteams * j) * 16;
            gather_tile(P.G[1], row0, tagB);
            __syncthreads();
            const f32x4 acc = gemm_phase(wb);
            if (kp == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hs[(4 * q + i) * 36 + 16 * nt + c] = gelu_f(acc[i] + bias[2]);
            }
            __syncthreads();
            if (wave == 0) {
                f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(&hs[c * 36 + 16 * g + 4 * q]);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) pa = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], bw4[4 * g + s4], pa, 0, 0, 0);
                }
                if (c < ad) {
                    fql_u64* d = P.Vg + ((size_t)mem * M + row0 + 4 * q) * 16 + c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) st_granule(d + (size_t)i * 16, tagC, pa[i]);
                }
            }
        }
    }
    // ---------------- final: member 0 folds the last partials, writes clip(a_n) and advances the team's epoch
    if (mem == 0) {
        for (int j = 0; j < ntile; ++j) {
            const int row0 = (team + P.nteams * j) * 16;
            const float sum = gather_head(row0, tag0 + 3 * P.flow_steps);
            const int r = tid >> 4, jj = tid & 15;
            if (jj < ad) stg(P.tgt + (size_t)(row0 + r) * ap + jj, clip1(eas[512 * j + r * 32 + jj] + (sum + ldg(P.b4 + jj)) * inv_steps));
        }
        if (tid == 0) __hip_atomic_store((FQL_GAS unsigned*)(P.epoch + team), ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Encoder head backward entry (utils/encoders.py:98, MLP with activate_final): dZ = (dXa[:, :n] (+ dXb[:, :n])) GELU'(z),
// dXa / dXb = the layer-0 input gradients of the module's MLP(s) ([M, ld], encoding in the first n columns)
struct EncDzArgs {
    const float *dxa, *dxb;  // dxb may be null
    const float* z;          // [M, n] pre-activation of the encoder's Dense
    float* dz;               // [M, n]
    int M, n, ld;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_enc_dz_kernel(const EncDzArgs P) {
    const int e = blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e >= P.M * P.n) return;
    const int r = e / P.n, j = e - r * P.n;
    float g = ldg(P.dxa + (size_t)r * P.ld + j);
    if (P.dxb) g += ldg(P.dxb + (size_t)r * P.ld + j);
    stg(P.dz + e, g * ldg(P.z + e));   // P.z holds GELU'(z) (GF_SAVE_Z)
}

struct AssembleArgs {
    const float *obs, *noise;  // noise null => RNG keyed by (key, seed)
    float* X;
    uint64_t key, seed;
    int n, n_pad, od, ad, inp;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_assemble_kernel(AssembleArgs P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= P.n_pad) return;
    for (int j = lane; j < P.inp; j += 64) {
        float v = 0.f;
        if (b < P.n) {
            if (j < P.od) v = P.obs[(size_t)b * P.od + j];
            else if (j < P.od + P.ad) {
                const int a = j - P.od;
                v = P.noise ? P.noise[(size_t)b * P.ad + a] : rng_normal(P.key, P.seed, 9u, (uint32_t)b, (uint32_t)a);
            }
        }
        P.X[(size_t)b * P.inp + j] = v;
    }
}
// out[n, ad] = src[n, 16][:, :ad]
__global__ void fql_extract_kernel(const float* src, float* out, int n, int ad, int ap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * ad) out[e] = src[(size_t)(e / ad) * ap + (e % ad)];
}
// ReplayBuffer.add_transition (utils/datasets.py:483-491): one row into the ring
__global__ void fql_dataset_add_kernel(float* obs, float* act, float* rew, float* mask, float* nobs,
                                       const float* row, int64_t pos, int od, int ad) {
    // row = [obs(od) | act(ad) | reward | mask | next_obs(od)]
    const int j = threadIdx.x;
    if (j < od) { obs[pos * od + j] = row[j]; nobs[pos * od + j] = row[od + ad + 2 + j]; }
    if (j < ad) act[pos * ad + j] = row[od + j];
    if (j == 0) { rew[pos] = row[od + ad]; mask[pos] = row[od + ad + 1]; }
}

// ------------------------------------------------------------------------------------------------
// One launch per level of the side lane: gemm64 tiles, then weight-gradient tiles, then LayerNorm-backward tiles, then
// the single-workgroup loss / metric tasks of that level.
// Every kernel boundary costs the whole chip ~3 us of launch / cache-flush time whichever stream it sits on
// (experiments/multi_chain.hip), so independent work of one level shares a launch even across kernel families.
// ------------------------------------------------------------------------------------------------
#ifndef FQL_SIDE_WAVES
#define FQL_SIDE_WAVES 3
#endif
#ifndef FQL_SIDE_VGPRS
#define FQL_SIDE_VGPRS 168
#endif
template <bool BIG, bool SPLIT = false>   // BIG: the launch contains 64 x 64 tile tasks (batches >= 1024); SPLIT: precision = 2 (bf16 x 3)
__device__ __forceinline__ void side_body(const GemmTask* __restrict__ gt, int ngt, const WgradTask* __restrict__ wt, int nwt, const LnBwdTask* __restrict__ lt,
                                          int nlt, int tile_w, int tile_l, const MiscTask* __restrict__ mt, int tile_m, int prio) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    if (b < tile_w) {
        gemm_tile_dispatch<BIG, SPLIT>(gt[find_task(gt, ngt, b)], lds);
    } else if (b < tile_l) {
        const int bid = b - tile_w;
        if (SPLIT) wgrad_split_body(wt[find_task(wt, nwt, bid)], bid, lds);
        else wgrad_body(wt[find_task(wt, nwt, bid)], bid, lds);
    } else if (b < tile_m) {
        const int bid = b - tile_l;
        const LnBwdTask& T = lt[find_task(lt, nlt, bid)];
        float (*red)[16][16] = reinterpret_cast<float (*)[16][16]>(lds);
        if (T.dq) lnbwd_body<true>(T, bid, red);
        else lnbwd_body<false>(T, bid, red);
    } else {
        const MiscTask& T = mt[b - tile_m];
        switch (T.kind) {
            case MISC_POSTOS: fql_post_onestep_body(T.po, lds); break;
            case MISC_LOSS_CRITIC: fql_loss_critic_body(T.lc, lds); break;
            case MISC_LOSS_Q: fql_loss_q_body(T.lq, lds); break;
            default: fql_loss_bc_body(T.lb, lds); break;
        }
    }
}
// two kernels: the register budget of the common one is set so that two of its workgroups and one 512-thread Euler-chain
// workgroup (2 x 104 registers per SIMD) fit a CU together; the 64 x 64 tile body needs twice that
__global__ __launch_bounds__(FQL_THREADS, FQL_SIDE_WAVES) void fql_side_kernel(
    const GemmTask* __restrict__ gt, int ngt, const WgradTask* __restrict__ wt, int nwt, const LnBwdTask* __restrict__ lt, int nlt, int tile_w, int tile_l,
    const MiscTask* __restrict__ mt, int tile_m, int prio, int tl) {
    tl_enter(tl);
    side_body<false>(gt, ngt, wt, nwt, lt, nlt, tile_w, tile_l, mt, tile_m, prio);
    tl_exit(tl);
}
// precision = 2: same launch contract, split-bf16 tile and weight-gradient bodies (32-row tiles only)
__global__ __launch_bounds__(FQL_THREADS, FQL_SIDE_WAVES) void fql_side_split_kernel(
    const GemmTask* __restrict__ gt, int ngt, const WgradTask* __restrict__ wt, int nwt, const LnBwdTask* __restrict__ lt, int nlt, int tile_w, int tile_l,
    const MiscTask* __restrict__ mt, int tile_m, int prio, int tl) {
    tl_enter(tl);
    side_body<false, true>(gt, ngt, wt, nwt, lt, nlt, tile_w, tile_l, mt, tile_m, prio);
    tl_exit(tl);
}
__global__ __launch_bounds__(FQL_THREADS, 2) void fql_side_big_kernel(
    const GemmTask* __restrict__ gt, int ngt, const WgradTask* __restrict__ wt, int nwt, const LnBwdTask* __restrict__ lt, int nlt, int tile_w, int tile_l,
    const MiscTask* __restrict__ mt, int tile_m, int prio, int tl) {
    tl_enter(tl);
    side_body<true>(gt, ngt, wt, nwt, lt, nlt, tile_w, tile_l, mt, tile_m, prio);
    tl_exit(tl);
}
