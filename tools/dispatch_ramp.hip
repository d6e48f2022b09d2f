// What stretches the dispatch ramp (first wave entry -> last wave entry) of a one-generation launch of 1024 x 256 threads?
// Knobs: LDS bytes per workgroup, VGPRs per wave, kernarg size.  Entry/exit by s_memrealtime (100 MHz, chip-wide).
// build: hipcc -O3 --offload-arch=gfx950 tools/dispatch_ramp.hip -o tools/dispatch_ramp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct Pad { float v[160]; };          // 640 bytes of by-value kernel arguments
template <int LDSF, int BIGV, bool PAD>
__global__ void __launch_bounds__(256) k(unsigned long long* stamps, float* out, Pad pad, int spin) {
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    if (LDSF > 0) { lds[threadIdx.x] = x; __syncthreads(); x = lds[(threadIdx.x + 1) & 255]; }
    if (BIGV) asm volatile("v_mov_b32 v120, %1\n v_add_f32 %0, %0, v120" : "+v"(x) : "v"(x) : "v120");
    if (PAD) x += pad.v[spin & 127];
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    out[w * 64 + lane] = x;
    if (lane == 0) { stamps[2 * w] = t0; stamps[2 * w + 1] = __builtin_amdgcn_s_memrealtime(); }
}
// a kernel with ~48 KB of code of which only a few instructions run (kernel code size / instruction cache as a suspect)
__global__ void __launch_bounds__(256) k_bigcode(unsigned long long* stamps, float* out, int spin, int never) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    if (never) {
#pragma unroll
        for (int i = 0; i < 6000; ++i) x = __builtin_fmaf(x, 1.0000001f + i * 1e-7f, 1e-9f * i);
    }
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    out[w * 64 + lane] = x;
    if (lane == 0) { stamps[2 * w] = t0; stamps[2 * w + 1] = __builtin_amdgcn_s_memrealtime(); }
}
void run_bigcode(unsigned long long* d, float* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) k_bigcode<<<1024, 256>>>(d, out, 200, 0);
    hipEventRecord(a);
    for (int i = 0; i < 500; ++i) k_bigcode<<<1024, 256>>>(d, out, 200, 0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long* h = (unsigned long long*)malloc(4096 * 16);
    hipMemcpy(h, d, 4096 * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, e1 = 0, x1 = 0;
    for (int w = 0; w < 4096; ++w) if (h[2 * w] < t0) t0 = h[2 * w];
    for (int w = 0; w < 4096; ++w) { if (h[2 * w] > e1) e1 = h[2 * w]; if (h[2 * w + 1] > x1) x1 = h[2 * w + 1]; }
    printf("%-44s %6.2f us/launch | last entry +%.2f us | last exit +%.2f us\n", "48 KB of code, few instructions executed", ms * 2.0f, (e1 - t0) / 100.0, (x1 - t0) / 100.0);
    free(h);
}
// true entry time vs the time the by-value kernel arguments (640 bytes, several cache lines) have arrived, per XCD
__global__ void __launch_bounds__(256) k_kernarg(unsigned long long* stamps, float* out, Pad pad, int spin) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float x = pad.v[0] + pad.v[40] + pad.v[80] + pad.v[120] + pad.v[159];          // five lines of the kernarg segment
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
    x += threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    out[w * 64 + lane] = x;
    if (lane == 0) { stamps[3 * w] = t0; stamps[3 * w + 1] = t1; stamps[3 * w + 2] = xcc; }
}
void run_kernarg(unsigned long long* d, float* out) {
    Pad pad = {};
    for (int i = 0; i < 50; ++i) k_kernarg<<<1024, 256>>>(d, out, pad, 200);
    hipDeviceSynchronize();
    k_kernarg<<<1024, 256>>>(d, out, pad, 200);
    hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(4096 * 24);
    hipMemcpy(h, d, 4096 * 24, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 4096; ++w) if (h[3 * w] < t0) t0 = h[3 * w];
    printf("by-value 640-byte kernarg: per XCD  first entry | last entry | first args-arrived | last args-arrived  (us after the first entry)\n");
    for (unsigned x = 0; x < 8; ++x) {
        double e0 = 1e9, e1 = 0, a0 = 1e9, a1 = 0;
        for (int w = 0; w < 4096; ++w) if (h[3 * w + 2] == x) {
            const double e = (h[3 * w] - t0) / 100.0, a = (h[3 * w + 1] - t0) / 100.0;
            if (e < e0) e0 = e; if (e > e1) e1 = e; if (a < a0) a0 = a; if (a > a1) a1 = a;
        }
        printf("  xcd %u: %5.2f | %5.2f | %5.2f | %5.2f\n", x, e0, e1, a0, a1);
    }
    free(h);
}
template <int LDSF, int BIGV, bool PAD> void run(const char* name, unsigned long long* d, float* out) {
    Pad pad = {};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) k<LDSF, BIGV, PAD><<<1024, 256>>>(d, out, pad, 200);
    hipEventRecord(a);
    for (int i = 0; i < 500; ++i) k<LDSF, BIGV, PAD><<<1024, 256>>>(d, out, pad, 200);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long* h = (unsigned long long*)malloc(4096 * 16);
    hipMemcpy(h, d, 4096 * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, e1 = 0, x1 = 0;
    for (int w = 0; w < 4096; ++w) if (h[2 * w] < t0) t0 = h[2 * w];
    for (int w = 0; w < 4096; ++w) { if (h[2 * w] > e1) e1 = h[2 * w]; if (h[2 * w + 1] > x1) x1 = h[2 * w + 1]; }
    printf("%-44s %6.2f us/launch | last entry +%.2f us | last exit +%.2f us\n", name, ms * 2.0f, (e1 - t0) / 100.0, (x1 - t0) / 100.0);
    free(h);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 24);
    float* out; hipMalloc(&out, 4096 * 64 * 4);
    run<0, 0, false>("no LDS, few VGPRs, small kernarg", d, out);
    run<8448, 0, false>("33 KB LDS per workgroup", d, out);
    run<0, 1, false>("121+ VGPRs", d, out);
    run<0, 0, true>("640-byte kernarg", d, out);
    run<8448, 1, true>("33 KB LDS + 121 VGPRs + 640-byte kernarg", d, out);
    run<2048, 0, false>("8 KB LDS per workgroup", d, out);
    run<8448, 1, false>("33 KB LDS + 121 VGPRs", d, out);
    run_bigcode(d, out);
    run_kernarg(d, out);
    return 0;
}
