// Kernels off the update's hot path, kept in a header of their own that is included LAST: code inserted in front of the GEMM-family
// kernels shifts their placement in the code object, which was measured to move the update rate by ~1 % (same box, A/B of builds).
#pragma once
#include "fql_kernels.h"

// ------------------------------------------------------------------------------------------------
// JAX-compatible noise on the device: the five tensors of one update from the five keys the reference's split
// chain derives (agents/fql.py:24,49-54,62-63,82,125,143-150; fql_amd/jax_prng.py restates the chain and both counter layouts).
// jax.random.normal = sqrt(2) erf_inv(uniform(nextafter(-1, 0), 1)), jax.random.uniform = (bits >> 9 | 0x3f800000) - 1 scaled;
// bits = Threefry-2x32 (20 rounds) in the original layout (counter array cut in two halves) or the partitionable one (counter =
// element index, output = word 0 ^ word 1).  erf_inv: the single-precision Giles polynomial XLA evaluates.
// ------------------------------------------------------------------------------------------------
struct JaxNoiseArgs {
    uint32_t key[5][2];   // eps1, x0, t, z, eps2
    float* out[5];
    int n[5];             // elements of each: B ad, B ad, B, B ad, B ad
    int partitionable;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    x0 += ks[0]; x1 += ks[1];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r0[4] = {13, 15, 26, 6}, r1[4] = {17, 29, 16, 24};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x0 += x1;
            x1 = rotl32(x1, (i & 1) ? r1[j] : r0[j]);
            x1 ^= x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
}
__device__ __forceinline__ float erfinv_giles(float x) {
    float w = -logf(__fmul_rn(1.0f - x, 1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}
__global__ __launch_bounds__(FQL_THREADS) void fql_jax_noise_kernel(const JaxNoiseArgs P) {
    const int which = blockIdx.y;
    int n = 0; float* out = nullptr; uint32_t k0 = 0, k1 = 0;
#pragma unroll
    for (int w = 0; w < 5; ++w) if (w == which) { n = P.n[w]; out = P.out[w]; k0 = P.key[w][0]; k1 = P.key[w][1]; }
    const int i = blockIdx.x * FQL_THREADS + threadIdx.x;
    if (i >= n) return;
    uint32_t x0, x1, bits;
    if (P.partitionable) {
        x0 = 0u; x1 = (uint32_t)i;
        threefry2x32(k0, k1, x0, x1);
        bits = x0 ^ x1;
    } else {
        const int half = (n + 1) >> 1;
        const int j = i < half ? i : i - half;
        x0 = (uint32_t)j; x1 = (j + half < n) ? (uint32_t)(j + half) : 0u;   // odd sizes: the counter array is padded with one zero
        threefry2x32(k0, k1, x0, x1);
        bits = i < half ? x0 : x1;
    }
    const float fl = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    if (which == 2) {                      // t ~ uniform(0, 1)
        out[i] = fmaxf(0.0f, fl);
    } else {                               // normal: minval = nextafter(-1, 0), maxval - minval rounds to 2 in float32
        const float lo = __uint_as_float(0xBF7FFFFFu);
        const float u = fmaxf(lo, __fadd_rn(__fmul_rn(fl, 2.0f), lo));
        out[i] = 1.41421354f * erfinv_giles(u);
    }
}

