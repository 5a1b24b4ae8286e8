// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) on Cody-Waite-reduced arguments
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = x[i];
    float rev = r * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    for (int i = 0; i < n; ++i) hx[i] = (float)((i + 0.5) / n * 2.0 - 1.0) * 0.7853981633974483f * (i % 7 == 0 ? 2.0f : 1.0f);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, es2 = 0, ec2 = 0;
    for (int i = 0; i < n; ++i) {
        double a = fabs((double)hs[i] - sin((double)hx[i])), b = fabs((double)hc[i] - cos((double)hx[i]));
        if (fabs(hx[i]) <= 0.7853981633974483) { if (a > es) es = a; if (b > ec) ec = b; }
        else { if (a > es2) es2 = a; if (b > ec2) ec2 = b; }
    }
    printf("|r|<=pi/4: max abs err sin %.3e cos %.3e ; pi/4<|r|<=pi/2: sin %.3e cos %.3e\n", es, ec, es2, ec2);
    return 0;
}
