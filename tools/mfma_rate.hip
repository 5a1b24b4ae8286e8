// f32 MFMA issue-rate probe: 1 vs 2 waves per SIMD, 32x32x2 vs 16x16x4, optional interleaved VALU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, float a0, float b0) {
    __shared__ float pad[MODE >= 100 ? 1 : 18432];  // MODE<100: 72 KB -> at most 2 blocks per CU
    if (threadIdx.x == 9999) pad[0] = 1;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f, v = a0;
    if (MODE % 100 == 0 || MODE % 100 == 2) {
        f32x16 c[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
                    if (MODE % 100 == 2) { v = v * 1.0001f + 0.5f; }
                }
        }
        float s = v;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 c[16] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
                    if (MODE % 100 == 3) { v = v * 1.0001f + 0.5f; }
                }
        }
        float s = v;
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += c[j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
template <int MODE>
void run(const char* name, int blocks, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 /*waves*/ * iters * (MODE % 2 == 0 ? 16 * 4096.0 : 32 * 2048.0);
    printf("%-44s blocks %4d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    run<0>("32x32x2, 1 wave/SIMD", 256, d);
    run<0>("32x32x2, 2 waves/SIMD", 512, d);
    run<1>("16x16x4, 1 wave/SIMD", 256, d);
    run<1>("16x16x4, 2 waves/SIMD", 512, d);
    run<2>("32x32x2 + 1 VALU fma per MFMA, 1 wave/SIMD", 256, d);
    run<2>("32x32x2 + 1 VALU fma per MFMA, 2 waves/SIMD", 512, d);
    run<3>("16x16x4 + 1 VALU fma per MFMA, 1 wave/SIMD", 256, d);
    run<3>("16x16x4 + 1 VALU fma per MFMA, 2 waves/SIMD", 512, d);
    return 0;
}
