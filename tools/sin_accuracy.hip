// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) against fp64, over [-4, 4] rad.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rev = x[i] * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = -4.0f + 8.0f * (float)i / (float)(n - 1);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
        es = fmax(es, fabs((double)s[i] - sin((double)x[i])));
        ec = fmax(ec, fabs((double)c[i] - cos((double)x[i])));
    }
    printf("v_sin_f32 max abs err %.3e   v_cos_f32 max abs err %.3e  (incl. the fp32 x/2pi product)\n", es, ec);
    return 0;
}
