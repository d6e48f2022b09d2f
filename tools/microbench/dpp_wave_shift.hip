#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    const int v = 100 + lane, old = -1000 - lane;
    out[lane] = __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false);         // wave_shl:1
    out[64 + lane] = __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);    // wave_shr:1
}
int main() {
    int* d; hipMalloc(&d, 128 * 4); int h[128];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shl:1  lane0 %d lane1 %d lane62 %d lane63 %d\n", h[0], h[1], h[62], h[63]);
    printf("wave_shr:1  lane0 %d lane1 %d lane62 %d lane63 %d\n", h[64], h[65], h[126], h[127]);
    return 0;
}
