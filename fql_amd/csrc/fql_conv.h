// Visual path: IMPALA encoder kernels for gfx950.
//   utils/encoders.py:10-58 (ResnetStack), :61-100 (ImpalaEncoder), utils/datasets.py:17-33,73-112 (frame stack, crop).
// All activations are NHWC fp32 (flax's layout), images arrive as uint8.  A 3x3 SAME convolution is an implicit GEMM on
// the fp32 matrix cores: rows = output pixels, columns = output channels (16 or 32), K = 9 taps x input channels.
#pragma once
#include "fql_kernels.h"

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
    const float* Wt;      // [9][Cw_rows][Cw_cols] arena layout of the forward kernel leaf (rows = fwd in-channels padded)
    const float* bias;    // [Co] or null
    float* out;           // [N,H,W,Co]
    float* out_relu;      // optional relu(out)
    const float* mask;    // optional, same shape as out: out *= (mask > 0)
    const float* add;     // optional, same shape as out: out += add
    int N, H, W;
    int Ci, Ci_real;      // staged input channels (multiple of 16) and channels present in memory
    int Co;               // output channels (16 or 32)
    int in_mode;          // 0 plain, 1 relu on load, 2 uint8 / 255
    int transposed;       // 0 forward, 1 data gradient (weights read as K[8-t][out][in])
    int Cw_rows, Cw_cols; // arena dims of the forward leaf per tap
    int R;                // image rows per workgroup
};

__device__ __forceinline__ int conv_lds_floats(int R, int W, int Ci, int Co) { return (R + 2) * (W + 2) * (Ci + 4) + Co * (9 * Ci + 4); }

template <int CO_TILES>
__device__ __forceinline__ void conv_body(const ConvArgs& P, float* lds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, q = lane >> 4;
    const int H = P.H, W = P.W, Ci = P.Ci, Co = P.Co, R = P.R;
    const int CS = Ci + 4, WS = 9 * Ci + 4, PW = W + 2;
    float* in_s = lds;                                 // [(R+2)][(W+2)][CS]
    float* w_s = lds + (R + 2) * PW * CS;              // [Co][WS]
    const int blocks_per_img = H / R;
    const int nblocks = P.N * blocks_per_img;

    // ---- weights -> LDS, transposed to [out channel][k]; staged once, the workgroup then walks its share of the row blocks
    {
        const int total = 9 * Ci * Co;
        for (int e = tid; e < total; e += FQL_THREADS) {
            const int o = e % Co, k = e / Co, t = k / Ci, ci = k - t * Ci;
            float v;
            if (!P.transposed) v = (ci < P.Cw_rows) ? ldg(P.Wt + ((size_t)t * P.Cw_rows + ci) * P.Cw_cols + o) : 0.f;
            else v = (o < P.Cw_rows) ? ldg(P.Wt + ((size_t)(8 - t) * P.Cw_rows + o) * P.Cw_cols + ci) : 0.f;
            w_s[o * WS + k] = v;
        }
    }
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int n = blk / blocks_per_img, y0 = (blk % blocks_per_img) * R;
    __syncthreads();  // the previous block's fragments are consumed (first pass: orders nothing that matters)
    // ---- input rows y0-1 .. y0+R with zero halo
    if (P.in_mode == 2) {
        const unsigned char* src = (const unsigned char*)P.in + (size_t)n * H * W * P.Ci_real;
        const int total = (R + 2) * PW * Ci;
        for (int e = tid; e < total; e += FQL_THREADS) {
            const int ci = e % Ci, px = e / Ci, xx = px % PW - 1, yy = y0 + px / PW - 1;
            float v = 0.f;
            if (ci < P.Ci_real && xx >= 0 && xx < W && yy >= 0 && yy < H) v = (float)src[((size_t)yy * W + xx) * P.Ci_real + ci] * (1.0f / 255.0f);
            in_s[px * CS + ci] = v;
        }
    } else {
        const float* src = (const float*)P.in + (size_t)n * H * W * Ci;
        const int c4 = Ci >> 2, total = (R + 2) * PW * c4;
        for (int e = tid; e < total; e += FQL_THREADS) {
            const int cc = e % c4, px = e / c4, xx = px % PW - 1, yy = y0 + px / PW - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                v = ldg4(src + ((size_t)yy * W + xx) * Ci + 4 * cc);
                if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            }
            *reinterpret_cast<f32x4*>(in_s + px * CS + 4 * cc) = v;
        }
    }
    __syncthreads();

    // ---- MFMA: this wave's row tiles are `wave` and `wave + 4` (16 consecutive pixels of the R x W block each)
    const int ntiles = R * W / 16;
    f32x4 acc[2][CO_TILES];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = min(wave + 4 * i, ntiles - 1);  // clamped: results of a duplicate tile are discarded
        const int p = 16 * t + c;
        pbase[i] = ((p / W) * PW + (p % W)) * CS + 4 * q;
    }
    const int ngroups = Ci >> 4;
    for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * PW + (t % 3)) * CS;
        for (int g = 0; g < ngroups; ++g) {
            f32x4 a[2], b[CO_TILES];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(in_s + pbase[i] + toff + 16 * g);
#pragma unroll
            for (int j = 0; j < CO_TILES; ++j) b[j] = *reinterpret_cast<const f32x4*>(w_s + (16 * j + c) * WS + t * Ci + 16 * g + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CO_TILES; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    }
    // ---- epilogue.  C layout: col = lane & 15 (channel), row = 4 q + r (pixel of the tile)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = wave + 4 * i;
        if (t >= ntiles) continue;
#pragma unroll
        for (int j = 0; j < CO_TILES; ++j) {
            const int ch = 16 * j + c;
            const float bv = P.bias ? ldg(P.bias + ch) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * t + 4 * q + r;
                const size_t o = (((size_t)n * H + y0 + p / W) * W + (p % W)) * Co + ch;
                float v = acc[i][j][r] + bv;
                if (P.mask) v = (ldg(P.mask + o) > 0.f) ? v : 0.f;
                if (P.add) v += ldg(P.add + o);
                stg(P.out + o, v);
                if (P.out_relu) stg(P.out_relu + o, fmaxf(v, 0.f));
            }
        }
    }
    }  // row blocks
}

__global__ __launch_bounds__(FQL_THREADS) void fql_conv3x3_kernel(const ConvArgs P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (P.Co == 32) conv_body<2>(P, lds);
    else conv_body<1>(P, lds);
}

// ------------------------------------------------------------------------------------------------
// conv3x3 weight gradient: dK[t][c][o] = sum_{n,y,x} f(in[n,y+ty-1,x+tx-1,c]) dOut[n,y,x,o], db[o] = sum dOut.
// Contraction over N H W pixels: every workgroup walks its share of the (image, row block) list with the input rows and
// the dOut rows in LDS.  The 9 (Ci/16) (tap, input-channel tile) units are dealt round-robin to the 4 waves; a wave keeps
// its units' 16 x Co accumulators in registers over ALL pixels the workgroup sees (no cross-wave reduction), so one
// partial [9 Ci + 1][Co] per workgroup goes to memory and fql_conv_wgrad_reduce_kernel folds them in fixed order.
// ------------------------------------------------------------------------------------------------
struct ConvWgradArgs {
    const void* in;      // forward input of the convolution ([N,H,W,Ci_real] float or uint8)
    const float* dout;   // [N,H,W,Co]
    float* partial;      // [gridDim.x][9 Ci + 1][Co]   (last row: bias partial)
    int N, H, W, Ci, Ci_real, Co, in_mode, R, nblocks;
};

template <int CI_TILES, int CO_TILES>
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

    for (int blk = blockIdx.x; blk < P.nblocks; blk += gridDim.x) {
        const int n = blk / blocks_per_img, y0 = (blk % blocks_per_img) * R;
        __syncthreads();  // previous block's fragments are consumed
        if (P.in_mode == 2) {
            const unsigned char* src = (const unsigned char*)P.in + (size_t)n * H * W * P.Ci_real;
            const int total = (R + 2) * PW * Ci;
            for (int e = tid; e < total; e += FQL_THREADS) {
                const int ci = e % Ci, px = e / Ci, xx = px % PW - 1, yy = y0 + px / PW - 1;
                float v = 0.f;
                if (ci < P.Ci_real && xx >= 0 && xx < W && yy >= 0 && yy < H) v = (float)src[((size_t)yy * W + xx) * P.Ci_real + ci] * (1.0f / 255.0f);
                in_s[px * CS + ci] = v;
            }
        } else {
            const float* src = (const float*)P.in + (size_t)n * H * W * Ci;
            const int c4 = Ci >> 2, total = (R + 2) * PW * c4;
            for (int e = tid; e < total; e += FQL_THREADS) {
                const int cc = e % c4, px = e / c4, xx = px % PW - 1, yy = y0 + px / PW - 1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    v = ldg4(src + ((size_t)yy * W + xx) * Ci + 4 * cc);
                    if (P.in_mode == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                }
                *reinterpret_cast<f32x4*>(in_s + px * CS + 4 * cc) = v;
            }
        }
        {
            const float* src = P.dout + ((size_t)n * H + y0) * W * Co;
            const int c4 = Co >> 2, total = R * W * c4;
            for (int e = tid; e < total; e += FQL_THREADS) {
                const int cc = e % c4, px = e / c4;
                *reinterpret_cast<f32x4*>(d_s + px * DS + 4 * cc) = ldg4(src + (size_t)px * Co + 4 * cc);
            }
        }
        __syncthreads();
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
            int pb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int p = 16 * pg + 4 * q + s;
                pb[s] = ((p / W) * PW + (p % W)) * CS;
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
    float* out = P.partial + (size_t)blockIdx.x * (size_t)(9 * Ci + 1) * Co;
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

__global__ __launch_bounds__(FQL_THREADS) void fql_conv_wgrad_kernel(const ConvWgradArgs P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (P.Ci == 32 && P.Co == 32) conv_wgrad_body<2, 2>(P, lds);
    else if (P.Ci == 16 && P.Co == 32) conv_wgrad_body<1, 2>(P, lds);
    else if (P.Ci == 32 && P.Co == 16) conv_wgrad_body<2, 1>(P, lds);
    else conv_wgrad_body<1, 1>(P, lds);
}

// dK (arena layout [9][Cw_rows][Co]) and db from the per-workgroup partials.  One workgroup = 64 consecutive elements; its
// 4 waves take every 4th partial each (coalesced 256-byte rows), meet in LDS and are added in wave order: deterministic.
struct ConvWredArgs {
    const float* partial;
    float* dK;
    float* db;
    int nparts, Ci, Co, Cw_rows;
};
__global__ __launch_bounds__(FQL_THREADS) void fql_conv_wgrad_reduce_kernel(const ConvWredArgs P) {
    __shared__ float red[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e = blockIdx.x * 64 + lane;
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
    const int k = e / P.Co, o = e - k * P.Co;
    if (k == 9 * P.Ci) { P.db[o] = s; return; }
    const int t = k / P.Ci, ci = k - t * P.Ci;
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
    const size_t e = (size_t)blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e >= (size_t)P.N * Ho * Wo * c4) return;
    const int cc = (int)(e % c4);
    size_t r = e / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
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
    const size_t e = (size_t)blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e >= (size_t)P.N * P.H * P.W * c4) return;
    const int cc = (int)(e % c4);
    size_t r = e / c4;
    const int x = (int)(r % P.W); r /= P.W;
    const int y = (int)(r % P.H);
    const int n = (int)(r / P.H);
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
};
__global__ __launch_bounds__(FQL_THREADS) void fql_img_gather_kernel(const ImgGatherArgs P) {
    const int KC = P.k * P.C;
    const size_t e = (size_t)blockIdx.x * FQL_THREADS + threadIdx.x;
    if (e >= (size_t)P.B * P.H * P.W * KC) return;
    const int ch = (int)(e % KC);
    size_t r = e / KC;
    const int x = (int)(r % P.W); r /= P.W;
    const int y = (int)(r % P.H);
    const int b = (int)(r / P.H);
    const int f = ch / P.C, c = ch - f * P.C;
    int sy = y, sx = x;
    if (P.crop) {
        sy = min(max(y + P.crop[2 * b] - P.pad, 0), P.H - 1);
        sx = min(max(x + P.crop[2 * b + 1] - P.pad, 0), P.W - 1);
    }
    const int64_t t = P.idx[b], i0 = P.init[b];
    const size_t pix = ((size_t)sy * P.W + sx) * P.C + c;
    const size_t img = (size_t)P.H * P.W * P.C;
    // frame f of obs is ob[max(t - (k-1-f), init)]; frame f of next is the same list shifted by one with next_ob[t] last
    const int64_t so = max(t - (P.k - 1 - f), i0);
    P.obs[e] = P.frames[(size_t)so * img + pix];
    if (f == P.k - 1) P.nobs[e] = P.next_frames[(size_t)t * img + pix];
    else P.nobs[e] = P.frames[(size_t)max(t - (P.k - 2 - f), i0) * img + pix];
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
};
__global__ __launch_bounds__(FQL_THREADS) void fql_img_index_kernel(const ImgIndexArgs P) {
    const int b = blockIdx.x * FQL_THREADS + threadIdx.x;
    if (b >= P.B) return;
    const uint64_t step = P.st->rng_step;
    int64_t i = P.idx_in ? P.idx_in[b] : P.lo + (int64_t)(((uint64_t)rng_u32(P.key, step, 7u, (uint32_t)b) * (uint64_t)P.span) >> 32);
    P.idx[b] = i;
    P.init[b] = P.ds_init[i];
    int cy = P.pad, cx = P.pad;   // crop_from == padding: the identity slice
    if (P.crop_in) { cy = P.crop_in[2 * b]; cx = P.crop_in[2 * b + 1]; }
    else if (P.p_aug > 0.f && rng_uniform(P.key, step, 8u, 0u) < P.p_aug) {   // ONE coin per batch (utils/datasets.py:90-92)
        const uint32_t w = 2u * (uint32_t)P.pad + 1u;
        cy = (int)(((uint64_t)rng_u32(P.key, step, 9u, (uint32_t)b) * w) >> 32);
        cx = (int)(((uint64_t)rng_u32(P.key, step, 10u, (uint32_t)b) * w) >> 32);
    }
    P.crop[2 * b] = cy; P.crop[2 * b + 1] = cx;
}
