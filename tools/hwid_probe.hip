// probe: which blocks share a CU and what HW_ID.wave_id do their waves get (2 blocks/CU via 72 KB LDS each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(256, 2) probe(unsigned* out, int spin) {
    __shared__ float big[18432];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    float v = big[threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.f) out[0] = 0;
}
int main() {
    const int blocks = 1024;
    unsigned* d; hipMalloc(&d, blocks * 8 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) {
        if (b < 24 || (b >= 256 && b < 272) || (b >= 512 && b < 528)) {
            printf("blk %4d:", b);
            for (int w = 0; w < 4; ++w) {
                unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1];
                printf("  [xcc %u se %u sh %u cu %2u simd %u wave %u]", xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
            }
            printf("\n");
        }
    }
    // count wave_id parity per (xcc,se,sh,cu) among first 512 blocks
    int odd = 0, even = 0;
    for (int b = 0; b < 512; ++b) { unsigned hw = h[(b * 4) * 2]; if (hw & 1) odd++; else even++; }
    printf("first 512 blocks: wave0 slot even %d odd %d\n", even, odd);
    return 0;
}
