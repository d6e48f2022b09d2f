// Tick rate of s_memtime (what the phase stamps record) against the constant 100 MHz s_memrealtime and against HIP events.
// build: hipcc -O2 --offload-arch=gfx950 tools/clock_calib.hip -o tools/clock_calib.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x * 1e-9f;
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)(x > 1e30f); }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 3 * 8 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {1, 1024}) for (int iters : {100000, 1000000}) {
        spin<<<blocks, 64>>>(d, iters); hipDeviceSynchronize();
        hipEventRecord(a); spin<<<blocks, 64>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %d iters %d: event %.1f us | s_memtime %llu ticks | s_memrealtime %llu ticks (100 MHz -> %.1f us) | memtime rate %.1f MHz | %.2f memtime ticks per dependent FMA\n",
               blocks, iters, ms * 1e3, h[0], h[1], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / iters);
    }
    return 0;
}
