// What bandwidth do the GEMM kernels' operand LOAD PATTERNS reach on their own (no MFMA, no LDS, no stores)?
//   pattern 0: param-grad style -- per K-tile a block reads 32 k-rows x 512 B of A and of B (row stride 2 KB), 1024 blocks
//   pattern 1: forward style    -- per K-tile a block reads 128 rows x 128 B of A (row stride 2 KB), 16384 blocks, 16 K-tiles
//   pattern 2: streaming        -- every wave reads consecutive 1-KB pieces
// build: hipcc --offload-arch=gfx950 -O3 -o tools/load_pattern tools/load_pattern.hip ; run: tools/load_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const float* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(unsigned)(bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : bytes), 0x00020000);
}

template <int PATTERN>
__global__ void __launch_bounds__(256, 2) k(unsigned* sink, const float* A, const float* B, long long n_rows, int ld, int ktiles) {
    const int tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    if (PATTERN == 0) {
        const long long k0 = (long long)(blockIdx.x >> 4) * ktiles * 32;   // split
        const int m0 = (blockIdx.x & 3) * 128, n0 = ((blockIdx.x >> 2) & 3) * 128;
        const __amdgpu_buffer_rsrc_t sa = srd(A + k0 * ld + m0, (long long)ktiles * 32 * ld * 4);
        const __amdgpu_buffer_rsrc_t sb = srd(B + k0 * ld + n0, (long long)ktiles * 32 * ld * 4);
        int v[4];
        for (int i = 0; i < 4; ++i) v[i] = ((4 * (tid >> 5) + i) * ld + (tid & 31) * 4) * 4;
        for (int t = 0; t < ktiles; ++t) {
            const int so = t * 32 * ld * 4;
            for (int i = 0; i < 4; ++i) {
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sa, v[i], so, 0);
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sb, v[i], so, 0);
            }
        }
    } else if (PATTERN == 1) {
        const long long m0 = (long long)(blockIdx.x >> 2) * 128;
        const __amdgpu_buffer_rsrc_t sa = srd(A + m0 * ld, 128ll * ld * 4);
        const __amdgpu_buffer_rsrc_t sb = srd(B + (long long)(blockIdx.x & 3) * 128 * ld, 128ll * ld * 4);
        int v[4];
        for (int i = 0; i < 4; ++i) v[i] = (((tid >> 2) + 64 * (i >> 1)) * ld + (tid & 3) * 8 + (i & 1) * 4) * 4;
        for (int t = 0; t < ktiles; ++t) {
            for (int i = 0; i < 4; ++i) {
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sa, v[i], t * 128, 0);
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sb, v[i], t * 128, 0);
            }
        }
    } else if (PATTERN == 3) {   // forward, blocked layout [panel][ktile][128][32]: one contiguous 16 KB chunk per K-tile
        const long long m0 = (long long)(blockIdx.x >> 2) * 128;
        const __amdgpu_buffer_rsrc_t sa = srd(A + m0 * ld, 128ll * ld * 4);
        const __amdgpu_buffer_rsrc_t sb = srd(B + (long long)(blockIdx.x & 3) * 128 * ld, 128ll * ld * 4);
        int v[4];
        for (int i = 0; i < 4; ++i) v[i] = (tid + 256 * i) * 16;
        for (int t = 0; t < ktiles; ++t) {
            for (int i = 0; i < 4; ++i) {
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sa, v[i], t * 16384, 0);
                acc ^= __builtin_amdgcn_raw_buffer_load_b128(sb, v[i], t * 16384, 0);
            }
        }
    } else if (PATTERN == 4) {   // param-grad, blocked operands: per K-tile 4 contiguous 4 KB chunks of A and of B
        const long long k0 = (long long)(blockIdx.x >> 4) * ktiles * 32;
        const int mt = blockIdx.x & 3, nt = (blockIdx.x >> 2) & 3;
        const __amdgpu_buffer_rsrc_t sa = srd(A, n_rows * ld * 4 > 0xFFFFFFFFll ? 0xFFFFFFFFll : n_rows * ld * 4);
        for (int t = 0; t < ktiles; ++t) {
            const long long row0 = k0 + t * 32;                        // 32 rows inside panel row0/128
            const long long panel = row0 >> 7, r_in = row0 & 127;
            for (int i = 0; i < 4; ++i) {                              // feature slice 4*mt + i of the panel, rows r_in..r_in+31
                const long long offa = ((panel * 16 + 4 * mt + i) * 128 + r_in) * 32 * 4 + tid * 16;
                const long long offb = ((panel * 16 + 4 * nt + i) * 128 + r_in) * 32 * 4 + tid * 16;
                acc ^= *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(A) + offa);
                acc ^= *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(B) + offb);
            }
        }
    } else {
        const u32x4* p = reinterpret_cast<const u32x4*>(A);
        const long long total = n_rows * ld / 4;
        for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) acc ^= p[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = acc[0];
}

int main() {
    const long long N = 524288; const int ld = 512;
    float *A, *B; unsigned* sink;
    hipMalloc(&A, N * ld * 4); hipMalloc(&B, N * ld * 4); hipMalloc(&sink, 4);
    hipMemset(A, 1, N * ld * 4); hipMemset(B, 2, N * ld * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, double bytes, const char* name) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    const int splits = 64, kt = (int)(N / 32 / splits);
    timeit([&] { hipLaunchKernelGGL(k<0>, dim3(16 * splits), dim3(256), 0, 0, sink, A, B, N, ld, kt); }, 2.0 * N * ld * 4, "param-grad pattern (2 GB)");
    timeit([&] { hipLaunchKernelGGL(k<1>, dim3((unsigned)(N / 128 * 4)), dim3(256), 0, 0, sink, A, B, N, ld, 16); }, 1.0 * N * ld * 4, "forward pattern (1 GB + L2)");
    timeit([&] { hipLaunchKernelGGL(k<3>, dim3((unsigned)(N / 128 * 4)), dim3(256), 0, 0, sink, A, B, N, ld, 16); }, 1.0 * N * ld * 4, "forward, blocked layout");
    timeit([&] { hipLaunchKernelGGL(k<4>, dim3(16 * splits), dim3(256), 0, 0, sink, A, B, N, ld, kt); }, 2.0 * N * ld * 4, "param-grad, blocked layout");
    timeit([&] { hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, sink, A, B, N, ld, 0); }, 1.0 * N * ld * 4, "streaming (1 GB)");
    return 0;
}
