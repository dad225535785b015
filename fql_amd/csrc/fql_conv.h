// Visual path: IMPALA encoder kernels for gfx950.
//   utils/encoders.py:10-58 (ResnetStack), :61-100 (ImpalaEncoder), utils/datasets.py:17-33,73-112 (frame stack, crop).
// All activations are NHWC fp32 (flax's layout), images arrive as uint8.  A 3x3 SAME convolution is an implicit GEMM on
// the fp32 matrix cores: rows = output pixels, columns = output channels (16 or 32), K = 9 taps x input channels.
#pragma once
#include "fql_kernels.h"
#ifndef FQL_CONV_WAVES
#define FQL_CONV_WAVES 3   // min waves per SIMD of the float convolution kernel (116 registers: up to 4)
#endif
#ifndef FQL_CONV_NS
#define FQL_CONV_NS 3   // float4 per thread of the input rows requested in one round, the rest in a rolled loop (6 cover every supported layer but cost a workgroup per CU of registers: measured slower)
#endif
#ifndef FQL_CWG_WAVES
#define FQL_CWG_WAVES 3   // min waves per SIMD of the convolution weight-gradient kernel: 165 registers, three workgroups per CU
#endif

// ------------------------------------------------------------------------------------------------
// conv3x3 (stride 1, zero padding 1), forward and data-gradient form.
//   forward:   out[n,y,x,o] = bias[o] + sum_{t,c} f(in[n,y+ty-1,x+tx-1,c]) K[t][c][o]          (utils/encoders.py:19-25)
//   transposed (dgrad): in = dOut, out[n,y,x,c] = sum_{t,o} in[n,y+ty-1,x+tx-1,o] K[8-t][c][o]   (jax.grad of the above)
// One workgroup = R image rows x W columns of one image (R W / 16 MFMA row tiles, <= 8, two per wave sharing every
// weight fragment).  The input rows (R + 2, zero padded halo) and ALL weights ([out channel][k], k = tap Ci + c) sit in
// LDS; both fragments are ds_read_b128 (lane (r, q) owns k = 16 g + 4 q + s, the k order is the same permutation for
// A and B so the product is unchanged).  Epilogue: + bias, x (mask > 0) (ReLU backward), + add (residual), and an
// optional second output relu(out) (the next consumer's input).
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const void* in;       // [N,H,W,Ci_real] float, or uint8 when in_mode == 2
    const float* Wl;      // weights already in the kernel's LDS layout [Co][9 Ci + 4] (fql_conv_wprep_kernel, once per pass)
    const float* bias;    // [Co] or null
    float* out;           // [N,H,W,Co]
    float* out_relu;      // optional relu(out)
    const float* mask;    // optional, same shape as out: out *= (mask > 0)
    const float* add;     // optional, same shape as out: out += add
    int N, H, W;
    int Ci, Ci_real;      // staged input channels (multiple of 16) and channels present in memory
    int Co;               // output channels (16 or 32)
    int in_mode;          // 0 plain, 1 relu on load, 2 uint8 / 255
    int transposed;       // 0 forward, 1 data gradient (informational: Wl is the matching layout)
    int R;                // image rows per workgroup
    int tile0, nwg;       // first workgroup of this task inside a shared launch, and how many it has
    unsigned char* parg;  // uint8 first convolution fused with the stack's max-pool (fql_conv3x3_pool_kernel): `out` is then the POOLED tensor
                          // [N, H/2, W/2, Co] and parg its argmax codes; the pre-pool tensor is never written
#ifdef FQL_STAMPS
    unsigned long long* stamps;   // diagnostics build (experiments/conv_bench.hip): [workgroup][8] wall-clock stamps of the first row block
#endif
};

// Input rows y0-1 .. y0+R of image n (zero halo) -> LDS [(R+2)][(W+2)][Ci+4], in two halves so the global loads of the NEXT
// row block fly while the matrix cores work on the current one: fetch() issues <= 6 unconditional 16-byte loads per thread
// into registers (clamped addresses, out-of-image lanes zeroed on commit), commit() writes them to LDS.
// uint8 images (first convolution): the rows are read as whole dwords (W C / 4 per row) and unpacked on commit; the halo
// columns and the channel padding are zeroed once per launch and never written again.
struct ConvTile {
    static constexpr int NF = 6;
    const void* in;
    float* in_s;
    int H, W, Ci, Ci_real, R, in_mode, CS, PW, tid;
    bool is_u8;   // uint8 input (in_mode == 2); a compile-time constant at every use, so the other staging path and its registers fold away
    static constexpr int NU8 = 4;   // dwords per thread on the uint8 path: (R + 2) W C / 4 / 256 <= 4 (W C <= 1152 at R = 1, 576 at R = 2)
    f32x4 pre[NF];
    unsigned prew[NU8];
    int u8off[NU8][4];   // LDS offsets of the 4 bytes of each of this thread's dwords: the same for every row block
    int f_lds[NF], f_g[NF], f_rr[NF];   // float path: LDS offset, offset inside an image row, tile row (-1: not this thread's); f_g < 0: halo column
    int y0;
    __device__ __forceinline__ void init() {
        if (is_u8) {
            const int total = (R + 2) * PW * CS;
            for (int e = tid; e < total; e += FQL_THREADS) in_s[e] = 0.f;
            const int dpr = (W * Ci_real) >> 2, tdw = (R + 2) * dpr;
#pragma unroll
            for (int i = 0; i < NU8; ++i) {
                const int e = min(tid + i * FQL_THREADS, tdw - 1);
                const int rr = e / dpr, cd = e - rr * dpr;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bi = 4 * cd + k, x = bi / Ci_real, ch = bi - x * Ci_real;
                    u8off[i][k] = (rr * PW + x + 1) * CS + ch;
                }
            }
        } else {
            const int c4 = Ci >> 2, total = (R + 2) * PW * c4;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int e = tid + i * FQL_THREADS;
                const int ec = min(e, total - 1);
                const int cc = ec % c4, px = ec / c4, xx = px % PW - 1;
                f_lds[i] = px * CS + 4 * cc;
                f_rr[i] = e < total ? px / PW : -1;
                f_g[i] = (xx >= 0 && xx < W) ? xx * Ci + 4 * cc : -1;
            }
        }
    }
    __device__ __forceinline__ void fetch(int n, int y0_) {
        y0 = y0_;
        if (is_u8) {
            const int dpr = (W * Ci_real) >> 2, total = (R + 2) * dpr;   // dwords per image row
            const FastDiv fdpr(dpr);
            const unsigned* src = (const unsigned*)((const unsigned char*)in + (size_t)n * H * W * Ci_real);
#pragma unroll
            for (int i = 0; i < NU8; ++i) {
                const int e = min(tid + i * FQL_THREADS, total - 1);
                int rr, cd;
                fdpr.divmod(e, rr, cd);
                const int yy = min(max(y0 + rr - 1, 0), H - 1);
                prew[i] = __builtin_nontemporal_load(src + (size_t)yy * dpr + cd);
            }
        } else {
            const float* src = (const float*)in + (size_t)n * H * W * Ci;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int yy = min(max(y0 + max(f_rr[i], 0) - 1, 0), H - 1);
                pre[i] = ldg4(src + (size_t)yy * W * Ci + max(f_g[i], 0));
            }
        }
    }
    __device__ __forceinline__ void commit() {
        if (is_u8) {
            const int dpr = (W * Ci_real) >> 2, total = (R + 2) * dpr;
            const FastDiv fdpr(dpr);
#pragma unroll
            for (int i = 0; i < NU8; ++i) {
                const int e = tid + i * FQL_THREADS;
                if (e >= total) continue;
                const int yy = y0 + fdpr.div(e) - 1;
                const bool ok = yy >= 0 && yy < H;
#pragma unroll
                for (int k = 0; k < 4; ++k) in_s[u8off[i][k]] = ok ? (float)((prew[i] >> (8 * k)) & 255u) * (1.0f / 255.0f) : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (f_rr[i] < 0) continue;
                const int yy = y0 + f_rr[i] - 1;
                f32x4 v = pre[i];
                if (in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (f_g[i] < 0 || yy < 0 || yy >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(in_s + f_lds[i]) = v;
            }
        }
    }
};

// PIPE: next block's input and this block's epilogue operands are fetched ahead of the MFMA loop (first convolution: uint8 rows,
// little else to hide the latency); the float layers run without it at a quarter of the registers and 2-3x the occupancy.
template <int CO_TILES, bool PIPE, bool SPLITR = false>   // SPLITR (precision = 2, uint8 first layer): fp32 LDS images, fragments split in registers, v_mfma_f32_16x16x16_bf16
__device__ __forceinline__ void conv_body(const ConvArgs& P, float* lds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, Ci = P.Ci, Co = P.Co, R = P.R;
    const int CS = Ci + 4, WS = 9 * Ci + 4, PW = W + 2;
    float* in_s = lds;                                 // [(R+2)][(W+2)][CS]
    float* w_s = lds + (R + 2) * PW * CS;              // [Co][WS]
    const int blocks_per_img = H / R;
    const int nblocks = P.N * blocks_per_img;
    const FastDiv fW(W), fPW(PW), fC4(Ci >> 2), fBPI(blocks_per_img);
    const int wg = (int)blockIdx.x - P.tile0, nwg = P.nwg;   // this task's workgroups walk its row blocks with stride nwg
#ifdef FQL_STAMPS
    unsigned long long stamp[8];
    int nst = 0;
#define VSTAMP() do { if (nst < 8) { __builtin_amdgcn_s_waitcnt(0); stamp[nst++] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define VSTAMP() do {} while (0)
#endif
    VSTAMP();   // [0] entry
    ConvTile T;
    T.in = P.in; T.in_s = in_s; T.H = H; T.W = W; T.Ci = Ci; T.Ci_real = P.Ci_real; T.R = R; T.CS = CS; T.PW = PW; T.tid = tid;
    T.in_mode = PIPE ? 2 : P.in_mode; T.is_u8 = PIPE;   // the pipelined body is the uint8 first layer only: a constant here lets the float staging path (and its registers) fold away
    if constexpr (PIPE) {
        T.init();
        int wn, wb;
        fBPI.divmod(wg, wn, wb);
        T.fetch(wn, wb * R);
    }

    // ---- weights -> LDS ([out channel][k]); staged once, the workgroup then walks its share of the row blocks
    {
        const int total = (Co * WS) >> 2;   // straight 16-byte copy: the layout was prepared by fql_conv_wprep_kernel
        for (int e = tid; e < total; e += FQL_THREADS) *reinterpret_cast<f32x4*>(w_s + 4 * e) = ldg4(P.Wl + 4 * e);
    }
    VSTAMP();   // [1] weights copied to LDS (stores issued)
    const int ntiles = R * W / 16;
    int pbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = min(wave + 4 * i, ntiles - 1);  // clamped: results of a duplicate tile are discarded
        const int p = 16 * t + c;
        int py, pxm;
        fW.divmod(p, py, pxm);
        pbase[i] = (py * PW + pxm) * CS + 4 * q;
    }
    // The product runs TRANSPOSED (weight fragment as the A operand, input fragment as B: the same registers, swapped), so lane (c, q) ends up with
    // output channels 16 j + 4 q .. + 3 of pixel c of a tile: bias, mask, residual and both outputs are 16-byte accesses, a wave's store is 1 KB contiguous
    // (a row block is R full image rows = R W consecutive pixels of the output tensor: no division anywhere in the epilogue).
    f32x4 bv[CO_TILES];
#pragma unroll
    for (int j = 0; j < CO_TILES; ++j) bv[j] = P.bias ? ldg4(P.bias + 16 * j + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int ngroups = Ci >> 4;

    for (int blk = wg; blk < nblocks; blk += nwg) {
        int n, y0;
        fBPI.divmod(blk, n, y0);
        y0 *= R;
        __syncthreads();  // the previous block's fragments are consumed (first pass: the zero fill / weights are in place)
        if constexpr (PIPE) {
            T.commit();
        } else {
            // staging: the first FQL_CONV_NS float4 of every thread are requested TOGETHER (clamped addresses, zeroed on commit) - one load round trip
            // instead of one per element: in-kernel stamps (experiments/conv_bench_st) showed the rolled loop at 4.1 us of a workgroup's 7.8 us, the MFMA
            // loop at 1.65 - then a rolled loop for what is left (none for the supported layers)
            const float* src = (const float*)P.in + (size_t)n * H * W * Ci;
            const int c4 = Ci >> 2, total = (R + 2) * PW * c4;
            {
                f32x4 sv[FQL_CONV_NS];
                int so[FQL_CONV_NS];
#pragma unroll
                for (int i = 0; i < FQL_CONV_NS; ++i) {
                    const int e = tid + i * FQL_THREADS, ec = min(e, total - 1);
                    int cc, px, prow, xx;
                    fC4.divmod(ec, px, cc);
                    fPW.divmod(px, prow, xx);
                    xx -= 1;
                    const int yy = y0 + prow - 1;
                    const bool inb = e < total && xx >= 0 && xx < W && yy >= 0 && yy < H;
                    sv[i] = ldg4(src + ((size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * Ci + 4 * cc);
                    so[i] = e < total ? ((px * CS + 4 * cc) << 1) | (inb ? 1 : 0) : -1;
                }
#pragma unroll
                for (int i = 0; i < FQL_CONV_NS; ++i) {
                    if (so[i] < 0) continue;
                    f32x4 v = sv[i];
                    if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (!(so[i] & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(in_s + (so[i] >> 1)) = v;
                }
            }
            for (int e = tid + FQL_CONV_NS * FQL_THREADS; e < total; e += FQL_THREADS) {
                int cc, px, prow, xx;
                fC4.divmod(e, px, cc);
                fPW.divmod(px, prow, xx);
                xx -= 1;
                const int yy = y0 + prow - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    v = ldg4(src + ((size_t)yy * W + xx) * Ci + 4 * cc);
                    if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                }
                *reinterpret_cast<f32x4*>(in_s + px * CS + 4 * cc) = v;
            }
        }
        __syncthreads();
        VSTAMP();   // [2] input rows staged
        const int nxt = blk + nwg;
        if constexpr (PIPE) { if (nxt < nblocks) { int nn, nb; fBPI.divmod(nxt, nn, nb); T.fetch(nn, nb * R); } }
        // epilogue operands of this block: issued now, consumed after the MFMA loop
        f32x4 mk[2][CO_TILES], ad[2][CO_TILES];
        if (PIPE)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = min(wave + 4 * i, ntiles - 1);
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) {
                const size_t o = ((((size_t)n * H + y0) * W) + 16 * t + c) * Co + 16 * j + 4 * q;
                mk[i][j] = P.mask ? ldg4(P.mask + o) : f32x4{1.f, 1.f, 1.f, 1.f};
                ad[i][j] = P.add ? ldg4(P.add + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 acc[2][CO_TILES];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * PW + (t % 3)) * CS;
            for (int g = 0; g < ngroups; ++g) {
                f32x4 a[2], b[CO_TILES];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(in_s + pbase[i] + toff + 16 * g);
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j) b[j] = *reinterpret_cast<const f32x4*>(w_s + (16 * j + c) * WS + t * Ci + 16 * g + 4 * q);
                if constexpr (SPLITR) {
                    u32x2 ah[2], al[2], bh[CO_TILES], bl[CO_TILES];
#pragma unroll
                    for (int i = 0; i < 2; ++i) bsplit4(a[i], ah[i], al[i]);
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) bsplit4(b[j], bh[j], bl[j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bh[j], al[i], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bl[j], ah[i], acc[i][j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bh[j], ah[i], acc[i][j]);
                } else
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][s], a[i][s], acc[i][j], 0, 0, 0);
            }
        }
        VSTAMP();   // [3] MFMA loop done
        // ---- epilogue.  C layout of the transposed product: col = lane & 15 = pixel of the tile, row = 4 q + r = channel of the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave + 4 * i;
            if (t >= ntiles) continue;
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) {
                const size_t o = ((((size_t)n * H + y0) * W) + 16 * t + c) * Co + 16 * j + 4 * q;
                f32x4 v = acc[i][j] + bv[j];
                if (PIPE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (mk[i][j][r] > 0.f) ? v[r] : 0.f;
                    v += ad[i][j];
                } else {
                    if (P.mask) {
                        const f32x4 m = ldg4(P.mask + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (m[r] > 0.f) ? v[r] : 0.f;
                    }
                    if (P.add) v += ldg4(P.add + o);
                }
                stg4(P.out + o, v);
                if (P.out_relu) stg4(P.out_relu + o, f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)});
            }
        }
        VSTAMP();   // [4] epilogue stored
    }  // row blocks
#ifdef FQL_STAMPS
    if (tid == 0 && P.stamps) {
        unsigned long long* d = P.stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < nst; ++i) d[i] = stamp[i];
    }
#endif
}

// One launch = every convolution of one scheduling level (e.g. the same layer of the four encoder passes): task table in HBM.
__global__ __launch_bounds__(FQL_THREADS, FQL_CONV_WAVES) void fql_conv3x3_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    if (P.Co == 32) conv_body<2, false>(P, lds);
    else conv_body<1, false>(P, lds);
}
__global__ __launch_bounds__(FQL_THREADS, 3) void fql_conv3x3_u8_split_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {   // in_mode 2, precision = 2
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    conv_body<1, true, true>(P, lds);
}
__global__ __launch_bounds__(FQL_THREADS, 3) void fql_conv3x3_u8_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {   // in_mode 2
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    conv_body<1, true>(P, lds);   // the first convolution of every supported encoder has 16 output channels (checked where the op is emitted)
}

// ------------------------------------------------------------------------------------------------
// uint8 first convolution + max_pool 3x3 / stride 2 / SAME of the same stack in ONE kernel (utils/encoders.py:19-33).  The pre-pool tensor
// (N H W 16 floats: 335 MB at B = 256 for the five encoder passes) was written once and read once by the pool kernel; here a workgroup
// computes the FIVE convolution rows 4 pr .. 4 pr + 4 two pooled rows need (one row of overlap between neighbouring workgroups: 25 % more
// convolution work, on a layer that is bound by its output write), parks them in LDS - the transposed product leaves a lane with four
// channels of a pixel, one 16-byte LDS store - and pools from there: window rows 2 oy .. 2 oy + 2, the -inf padding at the end, first maximum
// in window order (fql_maxpool_kernel's rule, bit for bit).  One workgroup per (image, pooled row pair); uint8 rows unpacked on the LDS store.
// SPLITR: precision = 2 (fragments split in registers, v_mfma_f32_16x16x16_bf16).
// ------------------------------------------------------------------------------------------------
// U8: uint8 input (first layer, 16 padded input channels, 16 output channels); else float NHWC input with CI channels (the first convolution of stacks 1, 2).
// NT: pixel tiles per wave = 5 W / 64 rounded up.
template <bool U8, int CI, int CO_TILES, bool SPLITR, int NT>
__device__ __forceinline__ void conv_pool_body(const ConvArgs& P, float* lds) {
    constexpr int Co = 16 * CO_TILES, CS = CI + 4, WS = 9 * CI + 4, OS = Co + 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, PW = W + 2;
    const int prs = (H + 3) >> 2;                        // pooled row pairs per image
    float* in_s = lds;                                   // [7][PW][CS]; reused as the convolution tile c_s [5][W][OS] once the MFMAs are done
    const int in_fl = 7 * PW * CS, cs_fl = 5 * W * OS;
    float* w_s = lds + (in_fl > cs_fl ? in_fl : cs_fl);  // [Co][WS]
    const int wg = (int)blockIdx.x - P.tile0;
    const FastDiv fPR(prs);
    int n, pr;
    fPR.divmod(wg, n, pr);
    const int y0 = 4 * pr;                               // first convolution row of this workgroup
    // ---- weights and input rows (y0 - 1 .. y0 + 5) -> LDS
    {
        const int total = (Co * WS) >> 2;
        for (int e = tid; e < total; e += FQL_THREADS) *reinterpret_cast<f32x4*>(w_s + 4 * e) = ldg4(P.Wl + 4 * e);
        if (U8) for (int e = tid; e < in_fl; e += FQL_THREADS) in_s[e] = 0.f;   // halo columns, channel padding, rows outside the image
    }
    if (U8) __syncthreads();
    if constexpr (U8) {
        const int Cr = P.Ci_real;
        const int dpr = (W * Cr) >> 2, total = 7 * dpr;  // dwords per image row
        const FastDiv fdpr(dpr), fCr(Cr);
        const unsigned* src = (const unsigned*)((const unsigned char*)P.in + (size_t)n * H * W * Cr);
        unsigned pw[4];
        int rr[4], cd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = min(tid + i * FQL_THREADS, total - 1);
            fdpr.divmod(e, rr[i], cd[i]);
            const int yy = min(max(y0 + rr[i] - 1, 0), H - 1);
            pw[i] = __builtin_nontemporal_load(src + (size_t)yy * dpr + cd[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * FQL_THREADS;
            const int yy = y0 + rr[i] - 1;
            if (e >= total || yy < 0 || yy >= H) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int x, ch;
                fCr.divmod(4 * cd[i] + k, x, ch);
                in_s[(rr[i] * PW + x + 1) * CS + ch] = (float)((pw[i] >> (8 * k)) & 255u) * (1.0f / 255.0f);
            }
        }
    } else {   // float rows: 7 PW CI / 4 float4 (<= 4 per thread for the supported layers, requested together), zero halo / outside rows
        const float* src = (const float*)P.in + (size_t)n * H * W * CI;
        constexpr int c4 = CI >> 2;
        const int total = 7 * PW * c4;
        const FastDiv fPW(PW);
        f32x4 sv[4];
        int so[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * FQL_THREADS, ec = min(e, total - 1);
            const int cc = ec % c4, px = ec / c4;
            int prow, xx;
            fPW.divmod(px, prow, xx);
            xx -= 1;
            const int yy = y0 + prow - 1;
            const bool inb = e < total && xx >= 0 && xx < W && yy >= 0 && yy < H;
            sv[i] = ldg4(src + ((size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * CI + 4 * cc);
            so[i] = e < total ? ((px * CS + 4 * cc) << 1) | (inb ? 1 : 0) : -1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (so[i] < 0) continue;
            f32x4 v = sv[i];
            if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (!(so[i] & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(in_s + (so[i] >> 1)) = v;
        }
    }
    __syncthreads();
    // ---- 5 rows x W pixels = 5 W / 16 pixel tiles, dealt to the four waves; a tile never crosses a row (W is a multiple of 16)
    const int ntiles = 5 * W / 16;
    f32x4 acc[NT][CO_TILES];
    int pbase[NT];
    const FastDiv fW(W);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = min(wave + 4 * i, ntiles - 1);
        int py, px;
        fW.divmod(16 * t + c, py, px);
        pbase[i] = (py * PW + px) * CS + 4 * q;
    }
    const int nmine = (ntiles - wave + 3) >> 2;          // tiles of this wave
    f32x4 bv[CO_TILES];
#pragma unroll
    for (int j = 0; j < CO_TILES; ++j) bv[j] = P.bias ? ldg4(P.bias + 16 * j + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * PW + (t % 3)) * CS;
#pragma unroll
        for (int g = 0; g < CI / 16; ++g) {
            f32x4 b[CO_TILES];
            u32x2 bh[CO_TILES], bl[CO_TILES];
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) {
                b[j] = *reinterpret_cast<const f32x4*>(w_s + (16 * j + c) * WS + t * CI + 16 * g + 4 * q);
                if constexpr (SPLITR) bsplit4(b[j], bh[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {   // (clamped duplicate tiles of the last waves are computed and dropped: every accumulator index stays static)
                const f32x4 a = *reinterpret_cast<const f32x4*>(in_s + pbase[i] + toff + 16 * g);
                if constexpr (SPLITR) {
                    u32x2 ah, al;
                    bsplit4(a, ah, al);
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) {
                        acc[i][j] = mfma_bf16_k16(bh[j], al, acc[i][j]);
                        acc[i][j] = mfma_bf16_k16(bl[j], ah, acc[i][j]);
                        acc[i][j] = mfma_bf16_k16(bh[j], ah, acc[i][j]);
                    }
                } else {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][s4], a[s4], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();   // every fragment read of in_s is done: the region becomes the convolution tile
    float* c_s = in_s;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = wave + 4 * i;
        if (i < nmine) {
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) *reinterpret_cast<f32x4*>(c_s + (16 * t + c) * OS + 16 * j + 4 * q) = acc[i][j] + bv[j];   // pixel 16 t + c, channels 16 j + 4 q ..
        }
    }
    __syncthreads();
    // ---- pool: 2 rows x W / 2 windows x Co / 4 channel quads, one float4 per thread and round
    const int Wo = W >> 1, Ho = H >> 1;
    constexpr int CQ = Co / 4;
    const int nout = 2 * Wo * CQ;
    for (int e = tid; e < nout; e += FQL_THREADS) {
        const int cq = e % CQ, r2 = e / CQ;
        const int orow = r2 >= Wo ? 1 : 0, ox = r2 - orow * Wo;
        const int oy = 2 * pr + orow;
        if (oy >= Ho) continue;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ly = 2 * orow + i, x = 2 * ox + j;      // local convolution row, column
                if (y0 + ly >= H || x >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(c_s + (ly * W + x) * OS + 4 * cq);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > best[k]) { best[k] = v[k]; arg[k] = 3 * i + j; }
            }
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * Co + 4 * cq;
        stg4(P.out + o, best);
        *reinterpret_cast<uchar4*>(P.parg + o) = make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]);
    }
}
// LDS floats of the fused kernels: max(7 (W + 2) (Ci + 4), 5 W (Co + 4)) + Co (9 Ci + 4)
#define FQL_CONV_POOL_LDS_FLOATS(W, Ci, Co) ((7 * ((W) + 2) * ((Ci) + 4) > 5 * (W) * ((Co) + 4) ? 7 * ((W) + 2) * ((Ci) + 4) : 5 * (W) * ((Co) + 4)) + (Co) * (9 * (Ci) + 4))
#define FQL_CONV_U8_POOL_LDS_FLOATS(W) FQL_CONV_POOL_LDS_FLOATS(W, 16, 16)
template <bool SPLITR>
__device__ __forceinline__ void conv_pool_dispatch(const ConvArgs& P, float* lds) {
    if (P.in_mode == 2) { if (P.W <= 64) conv_pool_body<true, 16, 1, SPLITR, 5>(P, lds); else conv_pool_body<true, 16, 1, SPLITR, 10>(P, lds); }
    else if (P.Ci == 16 && P.Co == 32) { if (P.W <= 32) conv_pool_body<false, 16, 2, SPLITR, 3>(P, lds); else conv_pool_body<false, 16, 2, SPLITR, 5>(P, lds); }
    else if (P.Ci == 32 && P.Co == 32) { if (P.W <= 32) conv_pool_body<false, 32, 2, SPLITR, 3>(P, lds); else conv_pool_body<false, 32, 2, SPLITR, 5>(P, lds); }
    else if (P.Ci == 16 && P.Co == 16) { if (P.W <= 32) conv_pool_body<false, 16, 1, SPLITR, 3>(P, lds); else conv_pool_body<false, 16, 1, SPLITR, 5>(P, lds); }
    else { if (P.W <= 32) conv_pool_body<false, 32, 1, SPLITR, 3>(P, lds); else conv_pool_body<false, 32, 1, SPLITR, 5>(P, lds); }
}
__global__ __launch_bounds__(FQL_THREADS, 3) void fql_conv3x3_pool_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_pool_dispatch<false>(tasks[find_task(tasks, ntasks, blockIdx.x)], lds);
}
__global__ __launch_bounds__(FQL_THREADS, 3) void fql_conv3x3_pool_split_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_pool_dispatch<true>(tasks[find_task(tasks, ntasks, blockIdx.x)], lds);
}

// ------------------------------------------------------------------------------------------------
// precision = 2: the float convolutions (forward and data gradient) with split-bf16 operands (fql_kernels.h, top).
// Same workgroup geometry, staging loop and epilogue as conv_body<., false>; the input rows are split ONCE by the thread that
// stages them into hi / lo planes [(R+2)][(W+2)][3 Ci / 4 words] (Ci / 2 words of data + padding that makes the fragment reads
// conflict-free: tools/lds_banks.py), the weights arrive pre-split from fql_conv_wprep_kernel as planes [Co][9 Ci / 2 + Ci / 4 words].
// Ci = 32: one tap is one 32-deep step (lane (r, q) owns channels 8q .. 8q+7: 16-byte fragment reads, v_mfma_f32_16x16x32_bf16);
// Ci = 16: one tap is one 16-deep step (channels 4q .. 4q+3: 8-byte reads, v_mfma_f32_16x16x16_bf16).  3 MFMAs per tap, row tile and
// channel tile instead of Ci / 4 fp32 ones.
// ------------------------------------------------------------------------------------------------
template <int CO_TILES, int CI>
__device__ __forceinline__ void conv_split_body(const ConvArgs& P, float* lds_f) {
    constexpr int PSW = 3 * CI / 4, WSW = 9 * CI / 2 + CI / 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, Co = P.Co, R = P.R;
    const int PW = W + 2;
    const int IPL = (R + 2) * PW * PSW, WPL = Co * WSW;
    unsigned* in_h = reinterpret_cast<unsigned*>(lds_f);   // [(R+2)][(W+2)][PSW]
    unsigned* in_l = in_h + IPL;
    unsigned* w_h = in_l + IPL;                             // [Co][WSW]
    unsigned* w_l = w_h + WPL;
    const int blocks_per_img = H / R;
    const int nblocks = P.N * blocks_per_img;
    const FastDiv fW(W), fPW(PW), fBPI(blocks_per_img);
    const int wg = (int)blockIdx.x - P.tile0, nwg = P.nwg;
    {   // weights -> LDS: hi plane then lo plane, a straight 16-byte copy (the padding words are zero in the copy)
        const int total = (2 * WPL) >> 2;
        const unsigned* src = reinterpret_cast<const unsigned*>(P.Wl);
        for (int e = tid; e < total; e += FQL_THREADS) *reinterpret_cast<u32x4*>(w_h + 4 * e) = ldg4u(src + 4 * e);
    }
    {   // the padding words of the input planes are never written by the staging loop: clear them once (the fragment reads do not touch them either; tidy)
        for (int e = tid; e < 2 * IPL; e += FQL_THREADS) in_h[e] = 0u;
    }
    const int ntiles = R * W / 16;
    constexpr int KW = CI == 32 ? 4 : 2;   // words per fragment
    int pbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = min(wave + 4 * i, ntiles - 1);  // clamped: results of a duplicate tile are discarded
        const int p = 16 * t + c;
        int py, pxm;
        fW.divmod(p, py, pxm);
        pbase[i] = (py * PW + pxm) * PSW + KW * q;
    }
    f32x4 bv[CO_TILES];   // transposed product (see conv_body): a lane owns channels 16 j + 4 q .. + 3 of pixel c
#pragma unroll
    for (int j = 0; j < CO_TILES; ++j) bv[j] = P.bias ? ldg4(P.bias + 16 * j + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    int wbase[CO_TILES];
#pragma unroll
    for (int j = 0; j < CO_TILES; ++j) wbase[j] = (16 * j + c) * WSW + KW * q;

    for (int blk = wg; blk < nblocks; blk += nwg) {
        int n, y0;
        fBPI.divmod(blk, n, y0);
        y0 *= R;
        __syncthreads();
        {
            const float* src = (const float*)P.in + (size_t)n * H * W * CI;
            constexpr int c4 = CI >> 2;
            const int total = (R + 2) * PW * c4;
            {   // one load round trip for the first FQL_CONV_NS float4 of every thread (see conv_body)
                f32x4 sv[FQL_CONV_NS];
                int so[FQL_CONV_NS];
#pragma unroll
                for (int i = 0; i < FQL_CONV_NS; ++i) {
                    const int e = tid + i * FQL_THREADS, ec = min(e, total - 1);
                    const int cc = ec % c4, px = ec / c4;   // c4 is a compile-time power of two
                    int prow, xx;
                    fPW.divmod(px, prow, xx);
                    xx -= 1;
                    const int yy = y0 + prow - 1;
                    const bool inb = e < total && xx >= 0 && xx < W && yy >= 0 && yy < H;
                    sv[i] = ldg4(src + ((size_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * CI + 4 * cc);
                    so[i] = e < total ? ((px * PSW + 2 * cc) << 1) | (inb ? 1 : 0) : -1;
                }
#pragma unroll
                for (int i = 0; i < FQL_CONV_NS; ++i) {
                    if (so[i] < 0) continue;
                    f32x4 v = sv[i];
                    if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (!(so[i] & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    u32x2 hi, lo;
                    bsplit4(v, hi, lo);
                    *reinterpret_cast<u32x2*>(in_h + (so[i] >> 1)) = hi;
                    *reinterpret_cast<u32x2*>(in_l + (so[i] >> 1)) = lo;
                }
            }
            for (int e = tid + FQL_CONV_NS * FQL_THREADS; e < total; e += FQL_THREADS) {
                const int cc = e % c4, px = e / c4;
                int prow, xx;
                fPW.divmod(px, prow, xx);
                xx -= 1;
                const int yy = y0 + prow - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    v = ldg4(src + ((size_t)yy * W + xx) * CI + 4 * cc);
                    if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                }
                u32x2 hi, lo;
                bsplit4(v, hi, lo);
                *reinterpret_cast<u32x2*>(in_h + px * PSW + 2 * cc) = hi;
                *reinterpret_cast<u32x2*>(in_l + px * PSW + 2 * cc) = lo;
            }
        }
        __syncthreads();
        f32x4 acc[2][CO_TILES];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * PW + (t % 3)) * PSW;
            if constexpr (CI == 32) {
                u32x4 ah[2], al[2], bh[CO_TILES], bl[CO_TILES];
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = *reinterpret_cast<const u32x4*>(in_h + pbase[i] + toff); al[i] = *reinterpret_cast<const u32x4*>(in_l + pbase[i] + toff); }
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j) { bh[j] = *reinterpret_cast<const u32x4*>(w_h + wbase[j] + 16 * t); bl[j] = *reinterpret_cast<const u32x4*>(w_l + wbase[j] + 16 * t); }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16(bh[j], al[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16(bl[j], ah[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16(bh[j], ah[i], acc[i][j]);
            } else {
                u32x2 ah[2], al[2], bh[CO_TILES], bl[CO_TILES];
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = *reinterpret_cast<const u32x2*>(in_h + pbase[i] + toff); al[i] = *reinterpret_cast<const u32x2*>(in_l + pbase[i] + toff); }
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j) { bh[j] = *reinterpret_cast<const u32x2*>(w_h + wbase[j] + 8 * t); bl[j] = *reinterpret_cast<const u32x2*>(w_l + wbase[j] + 8 * t); }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bh[j], al[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bl[j], ah[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = mfma_bf16_k16(bh[j], ah[i], acc[i][j]);
            }
        }
        // ---- epilogue (that of conv_body).  Transposed product: col = lane & 15 = pixel of the tile, row = 4 q + r = channel of the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = wave + 4 * i;
            if (t >= ntiles) continue;
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) {
                const size_t o = ((((size_t)n * H + y0) * W) + 16 * t + c) * Co + 16 * j + 4 * q;
                f32x4 v = acc[i][j] + bv[j];
                if (P.mask) {
                    const f32x4 m = ldg4(P.mask + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (m[r] > 0.f) ? v[r] : 0.f;
                }
                if (P.add) v += ldg4(P.add + o);
                stg4(P.out + o, v);
                if (P.out_relu) stg4(P.out_relu + o, f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)});
            }
        }
    }  // row blocks
}
#define FQL_CONV_SPLIT_LDS_WORDS(R, W, Ci, Co) (2 * ((R) + 2) * ((W) + 2) * (3 * (Ci) / 4) + 2 * (Co) * (9 * (Ci) / 2 + (Ci) / 4))
__global__ __launch_bounds__(FQL_THREADS, FQL_CONV_WAVES) void fql_conv3x3_split_kernel(const ConvArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    if (P.Ci == 32) { if (P.Co == 32) conv_split_body<2, 32>(P, lds); else conv_split_body<1, 32>(P, lds); }
    else { if (P.Co == 32) conv_split_body<2, 16>(P, lds); else conv_split_body<1, 16>(P, lds); }
}

// ------------------------------------------------------------------------------------------------
// conv3x3 weight gradient: dK[t][c][o] = sum_{n,y,x} f(in[n,y+ty-1,x+tx-1,c]) dOut[n,y,x,o], db[o] = sum dOut.
// Contraction over N H W pixels: every workgroup walks its share of the (image, row block) list with the input rows and
// the dOut rows in LDS (the next block's rows are already in flight, see ConvTile).  The 9 (Ci/16) (tap, input-channel
// tile) units are dealt round-robin to the 4 waves; a wave keeps its units' 16 x Co accumulators in registers over ALL
// pixels the workgroup sees (no cross-wave reduction), so one partial [9 Ci + 1][Co] per workgroup goes to memory and
// fql_conv_wgrad_reduce_kernel folds them in fixed order.
// ------------------------------------------------------------------------------------------------
struct ConvWgradArgs {
    const void* in;      // forward input of the convolution ([N,H,W,Ci_real] float or uint8)
    const float* dout;   // [N,H,W,Co]
    float* partial;      // [gridDim.x][9 Ci + 1][Co]   (last row: bias partial)
    int N, H, W, Ci, Ci_real, Co, in_mode, R, nblocks;
    int tile0, nwg;
};

// SPLIT (precision = 2, float inputs): the LDS images and the staging stay fp32; the 8 pixels a lane holds per TWO pixel groups for a unit's
// input fragment and for each dOut column tile are one 16x16x32 operand each (the k order inside a step is free as long as both operands
// share it), split in registers: 3 bf16 MFMAs per unit and column tile per 32 pixels instead of 8 fp32 ones.
template <int CI_TILES, int CO_TILES, bool U8, bool SPLIT = false>
__device__ __forceinline__ void conv_wgrad_body(const ConvWgradArgs& P, float* lds) {
    constexpr int NUNITS = 9 * CI_TILES, NU = (NUNITS + 3) / 4;  // units per wave (round-robin: unit = wave + 4 k)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, Ci = P.Ci, Co = P.Co, R = P.R;
    const int CS = Ci + 4, PW = W + 2, DS = Co + 4;
    float* in_s = lds;                       // [(R+2)][(W+2)][CS]
    float* d_s = lds + (R + 2) * PW * CS;    // [R W][DS]
    const int blocks_per_img = H / R;
    const int ntiles = R * W / 16;
    ConvTile T;
    T.in = P.in; T.in_s = in_s; T.H = H; T.W = W; T.Ci = Ci; T.Ci_real = P.Ci_real; T.R = R; T.in_mode = P.in_mode; T.CS = CS; T.PW = PW; T.tid = tid;
    T.is_u8 = U8;
    constexpr int ND = 4;   // dOut float4 per thread: R W Co / 4 / 256 <= 128 * 8 / 256
    f32x4 dpre[ND];
    const int dc4 = Co >> 2, dtotal = R * W * dc4;
    const FastDiv fW(W), fDC4(dc4), fBPI(blocks_per_img);
    auto fetch_d = [&](int n, int y0) {
        const float* src = P.dout + ((size_t)n * H + y0) * W * Co;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = min(tid + i * FQL_THREADS, dtotal - 1);
            int ep, ec;
            fDC4.divmod(e, ep, ec);
            dpre[i] = ldg4(src + (size_t)ep * Co + 4 * ec);
        }
    };
    auto commit_d = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + i * FQL_THREADS;
            if (e < dtotal) { int ep, ec; fDC4.divmod(e, ep, ec); *reinterpret_cast<f32x4*>(d_s + ep * DS + 4 * ec) = dpre[i]; }
        }
    };
    const int wg = (int)blockIdx.x - P.tile0, nwg = P.nwg;
    T.init();
    if (wg < P.nblocks) {
        int wn, wb;
        fBPI.divmod(wg, wn, wb);
        T.fetch(wn, wb * R);
        fetch_d(wn, wb * R);
    }
    f32x4 acc[NU][CO_TILES];
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[CO_TILES];
#pragma unroll
    for (int j = 0; j < CO_TILES; ++j) bs[j] = 0.f;
    int uoff[NU];  // LDS offset of unit k's tap and channel tile (CS-strided pixel layout)
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = min(wave + 4 * k, NUNITS - 1), t = u / CI_TILES, i = u - t * CI_TILES;
        uoff[k] = ((t / 3) * PW + (t % 3)) * CS + 16 * i + c;
    }

    for (int blk = wg; blk < P.nblocks; blk += nwg) {
        __syncthreads();  // previous block's fragments are consumed
        T.commit();
        commit_d();
        __syncthreads();
        const int nxt = blk + nwg;
        if (nxt < P.nblocks) {
            int nn, nb;
            fBPI.divmod(nxt, nn, nb);
            T.fetch(nn, nb * R);
            fetch_d(nn, nb * R);
        }
        if constexpr (SPLIT) {
            for (int pg = 0; pg < ntiles; pg += 2) {   // R W / 16 is even for every supported layer (checked where the op is emitted)
                float bf[CO_TILES][8];
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j)
#pragma unroll
                    for (int s = 0; s < 8; ++s) bf[j][s] = d_s[(16 * (pg + (s >> 2)) + 4 * q + (s & 3)) * DS + 16 * j + c];
                if (wave == 0) {
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) bs[j] += ((bf[j][0] + bf[j][1]) + (bf[j][2] + bf[j][3])) + ((bf[j][4] + bf[j][5]) + (bf[j][6] + bf[j][7]));
                }
                u32x4 bh[CO_TILES], bl[CO_TILES];
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j)
#pragma unroll
                    for (int w = 0; w < 4; ++w) { unsigned h, l; bsplit2(bf[j][2 * w], bf[j][2 * w + 1], h, l); bh[j][w] = h; bl[j][w] = l; }
                int pb2[2];   // a lane's 4 pixels of a group are consecutive in one image row (W is a multiple of 8): base + i CS
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int py, pxm;
                    fW.divmod(16 * (pg + h) + 4 * q, py, pxm);
                    pb2[h] = (py * PW + pxm) * CS;
                }
#pragma unroll
                for (int k = 0; k < NU; ++k) {
                    if (wave + 4 * k < NUNITS) {   // wave-uniform
                        u32x4 ah, al;
#pragma unroll
                        for (int w = 0; w < 4; ++w) { unsigned h, l; bsplit2(in_s[pb2[w >> 1] + (2 * (w & 1)) * CS + uoff[k]], in_s[pb2[w >> 1] + (2 * (w & 1) + 1) * CS + uoff[k]], h, l); ah[w] = h; al[w] = l; }
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[k][j] = mfma_bf16(al, bh[j], acc[k][j]);
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[k][j] = mfma_bf16(ah, bl[j], acc[k][j]);
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[k][j] = mfma_bf16(ah, bh[j], acc[k][j]);
                    }
                }
            }
        } else
        for (int pg = 0; pg < ntiles; ++pg) {
            // B fragments: dOut[pixel 16 pg + 4 q + s][16 j + c]   (every wave reads them: they are shared by all units)
            float b[CO_TILES][4];
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) b[j][s] = d_s[(16 * pg + 4 * q + s) * DS + 16 * j + c];
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < CO_TILES; ++j) bs[j] += (b[j][0] + b[j][1]) + (b[j][2] + b[j][3]);
            }
            int pb[4];   // a lane's four pixels are consecutive in one image row (W is a multiple of 4)
            {
                int py, pxm;
                fW.divmod(16 * pg + 4 * q, py, pxm);
#pragma unroll
                for (int s = 0; s < 4; ++s) pb[s] = (py * PW + pxm + s) * CS;
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                if (wave + 4 * k < NUNITS) {   // wave-uniform
                    float a[4];  // A[row = in channel 16 i + c][k = pixel]
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[s] = in_s[pb[s] + uoff[k]];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < CO_TILES; ++j) acc[k][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[j][s], acc[k][j], 0, 0, 0);
                }
            }
        }
    }
    // ---- per-workgroup partial: rows k = t Ci + 16 i + 4 q + r, cols 16 j + c
    float* out = P.partial + (size_t)wg * (size_t)(9 * Ci + 1) * Co;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = wave + 4 * k;
        if (u >= NUNITS) continue;
        const int t = u / CI_TILES, i = u - t * CI_TILES;
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg(out + (size_t)(t * Ci + 16 * i + 4 * q + r) * Co + 16 * j + c, acc[k][j][r]);
    }
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j) {
            float v = bs[j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (q == 0) stg(out + (size_t)(9 * Ci) * Co + 16 * j + c, v);
        }
    }
}

__global__ __launch_bounds__(FQL_THREADS, FQL_CWG_WAVES) void fql_conv_wgrad_split_kernel(const ConvWgradArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvWgradArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    const bool even = ((P.R * P.W / 16) & 1) == 0;   // the split loop walks pixel groups in pairs
    if (P.in_mode == 2) { if (even) conv_wgrad_body<1, 1, true, true>(P, lds); else conv_wgrad_body<1, 1, true>(P, lds); }   // uint8 first layer
    else if (!even) {
        if (P.Ci == 32 && P.Co == 32) conv_wgrad_body<2, 2, false>(P, lds);
        else if (P.Ci == 16 && P.Co == 32) conv_wgrad_body<1, 2, false>(P, lds);
        else if (P.Ci == 32 && P.Co == 16) conv_wgrad_body<2, 1, false>(P, lds);
        else conv_wgrad_body<1, 1, false>(P, lds);
    }
    else if (P.Ci == 32 && P.Co == 32) conv_wgrad_body<2, 2, false, true>(P, lds);
    else if (P.Ci == 16 && P.Co == 32) conv_wgrad_body<1, 2, false, true>(P, lds);
    else if (P.Ci == 32 && P.Co == 16) conv_wgrad_body<2, 1, false, true>(P, lds);
    else conv_wgrad_body<1, 1, false, true>(P, lds);
}
__global__ __launch_bounds__(FQL_THREADS, FQL_CWG_WAVES) void fql_conv_wgrad_kernel(const ConvWgradArgs* __restrict__ tasks, int ntasks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvWgradArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    if (P.in_mode == 2) conv_wgrad_body<1, 1, true>(P, lds);   // the uint8 first layer (<= 16 channels in, 16 out: checked at emit)
    else if (P.Ci == 32 && P.Co == 32) conv_wgrad_body<2, 2, false>(P, lds);
    else if (P.Ci == 16 && P.Co == 32) conv_wgrad_body<1, 2, false>(P, lds);
    else if (P.Ci == 32 && P.Co == 16) conv_wgrad_body<2, 1, false>(P, lds);
    else conv_wgrad_body<1, 1, false>(P, lds);
}

// dK (arena layout [9][Cw_rows][Co]) and db from the per-workgroup partials.  One workgroup = 64 consecutive elements; its
// 4 waves take every 4th partial each (coalesced 256-byte rows), meet in LDS and are added in wave order: deterministic.
struct ConvWredArgs {
    const float* partial;
    float* dK;
    float* db;
    int nparts, Ci, Co, Cw_rows;
    int tile0;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_conv_wgrad_reduce_kernel(const ConvWredArgs* __restrict__ tasks, int ntasks) {
    __shared__ float red[4][64];
    const ConvWredArgs& P = tasks[find_task(tasks, ntasks, blockIdx.x)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = ((int)blockIdx.x - P.tile0) * 64 + lane;
    const int rows = 9 * P.Ci + 1, total = rows * P.Co;
    const size_t stride = (size_t)total;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int p = wave;
        for (; p + 4 < P.nparts; p += 8) {
            s0 += ldg(P.partial + (size_t)p * stride + e);
            s1 += ldg(P.partial + (size_t)(p + 4) * stride + e);
        }
        if (p < P.nparts) s0 += ldg(P.partial + (size_t)p * stride + e);
    }
    red[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave != 0 || e >= total) return;
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    int k, o, t, ci;
    FastDiv(P.Co).divmod(e, k, o);
    if (k == 9 * P.Ci) { P.db[o] = s; return; }
    FastDiv(P.Ci).divmod(k, t, ci);
    if (ci < P.Cw_rows) P.dK[((size_t)t * P.Cw_rows + ci) * P.Co + o] = s;
}

// ------------------------------------------------------------------------------------------------
// max_pool 3x3 / stride 2 / SAME (utils/encoders.py:27-33): window o covers rows 2o..2o+2 (the -inf pad is at the end).
// arg = 3 i + j of the first maximum in row-major window order.  4 channels per thread.
// ------------------------------------------------------------------------------------------------
struct PoolArgs {
    const float* in;     // [N,H,W,C]
    float* out;          // [N,H/2,W/2,C]
    unsigned char* arg;  // [N,H/2,W/2,C]
    int N, H, W, C;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_maxpool_kernel(const PoolArgs P) {
    const int c4 = P.C >> 2, Ho = P.H >> 1, Wo = P.W >> 1;
    const size_t e64 = (size_t)blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e64 >= (size_t)P.N * Ho * Wo * c4) return;   // (< 2^31 elements: checked where the op is emitted)
    if (e64 >> 31) return;   // (flat indices stay below 2^31: checked where the op is emitted)
    const FastDiv fC4(c4, e64), fWo(Wo, e64), fHo(Ho, e64);
    int cc, r, ox, oy, n;
    fC4.divmod((int)e64, r, cc);
    fWo.divmod(r, r, ox);
    fHo.divmod(r, n, oy);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int y = 2 * oy + i, x = 2 * ox + j;
            if (y >= P.H || x >= P.W) continue;
            const f32x4 v = ldg4(P.in + (((size_t)n * P.H + y) * P.W + x) * P.C + 4 * cc);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[k] > best[k]) { best[k] = v[k]; arg[k] = 3 * i + j; }
        }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * P.C + 4 * cc;
    stg4(P.out + o, best);
    *reinterpret_cast<uchar4*>(P.arg + o) = make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]);
}
// backward as a gather: input pixel (y, x) collects from the <= 4 windows that contain it and chose it
struct PoolBwdArgs {
    const float* dout;         // [N,H/2,W/2,C]
    const unsigned char* arg;
    float* din;                // [N,H,W,C]
    int N, H, W, C;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_maxpool_bwd_kernel(const PoolBwdArgs P) {
    const int c4 = P.C >> 2, Ho = P.H >> 1, Wo = P.W >> 1;
    const size_t e64 = (size_t)blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e64 >= (size_t)P.N * P.H * P.W * c4) return;   // (< 2^31 elements)
    if (e64 >> 31) return;
    const FastDiv fC4(c4, e64), fW(P.W, e64), fH(P.H, e64);
    int cc, r, x, y, n;
    fC4.divmod((int)e64, r, cc);
    fW.divmod(r, r, x);
    fH.divmod(r, n, y);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    // windows oy with 2 oy <= y <= 2 oy + 2
    for (int oy = max(0, (y - 1) >> 1); oy <= min(Ho - 1, y >> 1); ++oy)
        for (int ox = max(0, (x - 1) >> 1); ox <= min(Wo - 1, x >> 1); ++ox) {
            const int code = 3 * (y - 2 * oy) + (x - 2 * ox);
            const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * P.C + 4 * cc;
            const uchar4 a = *reinterpret_cast<const uchar4*>(P.arg + o);
            const f32x4 d = ldg4(P.dout + o);
            if (a.x == code) g[0] += d[0];
            if (a.y == code) g[1] += d[1];
            if (a.z == code) g[2] += d[2];
            if (a.w == code) g[3] += d[3];
        }
    stg4(P.din + (((size_t)n * P.H + y) * P.W + x) * P.C + 4 * cc, g);
}

// ------------------------------------------------------------------------------------------------
// Dataset side of the visual path (utils/datasets.py:73-112): frame stacking clamped to the episode start and the
// edge-padded random crop, fused into ONE gather that writes the uint8 [B,H,W,k C] batch the encoders read.
//   obs  = [ob[max(t-k+1, init)], ..., ob[t]]          next = [ob[max(t-k+2, init)], ..., ob[t], next_ob[t]]
//   crop: out[y][x] = in[clamp(y + cy - pad)][clamp(x + cx - pad)]  (edge padding + slice), same (cy, cx) for both.
// ------------------------------------------------------------------------------------------------
struct ImgGatherArgs {
    const unsigned char* frames;       // [Nrows,H,W,C]
    const unsigned char* next_frames;  // [Nrows,H,W,C]
    const int64_t* idx;                // [B] transition indices
    const int64_t* init;               // [B] first index of each transition's episode
    const int* crop;                   // [B][2] (cy, cx) or null (no augmentation)
    unsigned char* obs;                // [B,H,W,k C]
    unsigned char* nobs;
    int B, H, W, C, k, pad;
    // balanced sampling (main.py:255-259): rows [split, B) read the replay ring's frames; 0 = one source
    int split;
    const unsigned char* frames2;
    const unsigned char* next_frames2;
};
// One thread per FOUR output bytes of obs / nobs (H W k C is a multiple of 4: checked at upload), 32-bit indices with multiply-high division
// (the first version - one thread per byte, four 64-bit divisions each - took 53 us per update).
__global__ __launch_bounds__(FQL_THREADS) void fql_img_gather_kernel(const ImgGatherArgs P) {
    const int KC = P.k * P.C, rowb = P.W * KC;               // bytes per output pixel / image row
    const unsigned e4 = blockIdx.x * FQL_THREADS + threadIdx.x;
    const unsigned total4 = (unsigned)(((size_t)P.B * P.H * rowb) >> 2);
    if (e4 >= total4) return;
    if (e4 >> 31) return;   // (B H W k C < 2^33 bytes)
    const FastDiv fRow4(rowb >> 2, e4), fH(P.H, e4), fKC(KC), fC(P.C);
    int r, d4, b, y;
    fRow4.divmod((int)e4, r, d4);                             // r = image row index b H + y, d4 = dword inside the row
    fH.divmod(r, b, y);
    int sy = y, dx = 0;
    if (P.crop) {
        sy = min(max(y + P.crop[2 * b] - P.pad, 0), P.H - 1);
        dx = P.crop[2 * b + 1] - P.pad;
    }
    const int64_t t = P.idx[b], i0 = P.init[b];
    const size_t img = (size_t)P.H * P.W * P.C;
    const bool second = P.split > 0 && b >= P.split;
    const unsigned char* fr = second ? P.frames2 : P.frames;
    const unsigned char* nf = second ? P.next_frames2 : P.next_frames;
    unsigned wo = 0, wn = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x, ch, f, c;
        fKC.divmod(4 * d4 + i, x, ch);
        fC.divmod(ch, f, c);
        const int sx = min(max(x + dx, 0), P.W - 1);
        const size_t pix = ((size_t)sy * P.W + sx) * P.C + c;
        // frame f of obs is ob[max(t - (k-1-f), init)]; frame f of next is the same list shifted by one with next_ob[t] last
        const int64_t so = max(t - (P.k - 1 - f), i0);
        const unsigned vo = fr[(size_t)so * img + pix];
        const unsigned vn = (f == P.k - 1) ? nf[(size_t)t * img + pix] : fr[(size_t)max(t - (P.k - 2 - f), i0) * img + pix];
        wo |= vo << (8 * i);
        wn |= vn << (8 * i);
    }
    reinterpret_cast<unsigned*>(P.obs)[e4] = wo;
    reinterpret_cast<unsigned*>(P.nobs)[e4] = wn;
}

// index draw (utils/datasets.py:64-66), episode starts and crop offsets (utils/datasets.py:102-112) of one batch
struct ImgIndexArgs {
    const int64_t* idx_in;    // [B] or null = uniform in [lo, lo + span) from the engine RNG (same stream as the state path)
    const int* crop_in;       // [B][2] or null = engine RNG
    const int64_t* ds_init;   // [n] first index of each row's episode
    const DevState* st;
    uint64_t key;
    int64_t lo, span;
    float p_aug;
    int B, pad;
    int64_t* idx;             // out [B]
    int64_t* init;            // out [B]
    int* crop;                // out [B][2]
    // balanced sampling: rows [split, B) index the replay ring (its own range, episode starts and augmentation coin:
    // train_dataset.sample and replay_buffer.sample are two calls, main.py:257-258); 0 = one source
    int split;
    const int64_t* ds_init2;
    int64_t lo2, span2;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_img_index_kernel(const ImgIndexArgs P) {
    const int b = blockIdx.x * FQL_THREADS + threadIdx.x;
    if (b >= P.B) return;
    const uint64_t step = P.st->rng_step;
    const uint64_t key = P.key ^ (P.st->rng_stream * 0x9E3779B97F4A7C15ull);
    const bool second = P.split > 0 && b >= P.split;
    const uint64_t u = rng_u32(key, step, 7u, (uint32_t)b);
    int64_t i = P.idx_in ? P.idx_in[b] : (second ? P.lo2 + (int64_t)((u * (uint64_t)P.span2) >> 32) : P.lo + (int64_t)((u * (uint64_t)P.span) >> 32));
    P.idx[b] = i;
    P.init[b] = (second ? P.ds_init2 : P.ds_init)[i];
    int cy = P.pad, cx = P.pad;   // crop_from == padding: the identity slice
    if (P.crop_in) { cy = P.crop_in[2 * b]; cx = P.crop_in[2 * b + 1]; }
    else if (P.p_aug > 0.f && rng_uniform(key, step, 8u, second ? 1u : 0u) < P.p_aug) {   // ONE coin per sample() call (utils/datasets.py:90-92)
        const uint32_t w = 2u * (uint32_t)P.pad + 1u;
        cy = (int)(((uint64_t)rng_u32(key, step, 9u, (uint32_t)b) * w) >> 32);
        cx = (int)(((uint64_t)rng_u32(key, step, 10u, (uint32_t)b) * w) >> 32);
    }
    P.crop[2 * b] = cy; P.crop[2 * b + 1] = cx;
}

// Convolution weights from the arena leaf ([9][cin][cout], flax HWIO) into the two LDS layouts the conv kernel copies verbatim:
//   forward  Wf[o][t Ci + c]      = K[t][c][o]        (Ci = cin padded to 16, pad columns stay zero)
//   dgrad    Wb[c][t cout + o]    = K[8 - t][c][o]    (the flipped, transposed kernel of jax.grad's input gradient)
// Row strides 9 Ci + 4 / 9 cout + 4.  One launch covers every convolution of an encoder (blockIdx.y = convolution).
struct ConvWprepTask {
    const float* K;
    float* Wf;
    float* Wb;   // null: no data gradient needed (first convolution)
    int cin, cout, Ci;
    int split;   // precision = 2, float convolutions: Wf / Wb are bf16 hi / lo planes [rows][9 C / 2 + C / 4 words] (C = channels per tap) for
                 // fql_conv3x3_split_kernel instead of the fp32 [rows][9 C + 4] image (the uint8 first layer keeps the fp32 image).
                 // 1: both copies split; 2: only Wb (the forward copy stays fp32: a stack's first convolution fused with its max-pool splits in registers)
};
__device__ __forceinline__ void wprep_store_split(float* planes, int rows, int C, int row, int k, float v) {
    const int WSW = 9 * C / 2 + C / 4;
    unsigned h, l;
    bsplit2(v, 0.f, h, l);
    unsigned short* hp = reinterpret_cast<unsigned short*>(planes);
    unsigned short* lp = hp + (size_t)2 * rows * WSW;
    hp[(size_t)2 * row * WSW + k] = (unsigned short)(h & 0xFFFFu);
    lp[(size_t)2 * row * WSW + k] = (unsigned short)(l & 0xFFFFu);
}
__global__ __launch_bounds__(FQL_THREADS) void fql_conv_wprep_kernel(const ConvWprepTask* __restrict__ tasks) {
    const ConvWprepTask T = tasks[blockIdx.y];
    const int total = 9 * T.cin * T.cout;
    const int WSf = 9 * T.Ci + 4, WSb = 9 * T.cout + 4;
    for (int e = blockIdx.x * FQL_THREADS + threadIdx.x; e < total; e += gridDim.x * FQL_THREADS) {
        const int o = e % T.cout, r = e / T.cout, c = r % T.cin, t = r / T.cin;
        const float v = T.K[e];
        if (T.split) {
            if (T.split == 1) wprep_store_split(T.Wf, T.cout, T.Ci, o, t * T.Ci + c, v);
            else T.Wf[(size_t)o * WSf + t * T.Ci + c] = v;
            if (T.Wb) wprep_store_split(T.Wb, T.cin, T.cout, c, (8 - t) * T.cout + o, v);
            continue;
        }
        T.Wf[(size_t)o * WSf + t * T.Ci + c] = v;
        if (T.Wb) T.Wb[(size_t)c * WSb + (8 - t) * T.cout + o] = v;
    }
}
