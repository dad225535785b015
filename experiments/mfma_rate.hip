// Issue rate of the bf16 MFMA shapes the split kernels use, one wave per SIMD: cycles per instruction (s_memtime), 4 independent accumulators.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o experiments/mfma_rate experiments/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float x = (float)threadIdx.x * 1e-3f;
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(x + i); b8[i] = (__bf16)(x - i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x + u, acc[u], 0, 0, 0);
            if (KIND == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u], 0, 0, 0);
            if (KIND == 2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[u], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int iters = 4096;
    const char* names[3] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x32_bf16"};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.2f s_memtime ticks per instruction (one wave per SIMD, 4 accumulators)\n", names[kind], (double)c / (4.0 * iters));
    }
    return 0;
}
