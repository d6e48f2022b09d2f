// I/O skeleton of the headline kernel: the same grid (1024 workgroups x 4 wavefronts, one wavefront = 64 samples), the same
// HBM traffic (read q 28 B, write link positions 132 B + cost 4 B + gradient 28 B per sample = 192 B x 262144 = 50.3 MB)
// and the same access pattern (16-byte lanes, contiguous runs per wavefront), with NO arithmetic -- the floor the fused kernel
// can approach on this chip, launch overhead included.  Variants: store modifier, extra dependent FMAs per lane before the stores.
// build: hipcc -O3 --offload-arch=gfx950 tools/io_skeleton.hip -o tools/io_skeleton.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MOD> __device__ __forceinline__ void st(f4* p, f4 v) {
    if (MOD == 0) *p = v;
    else if (MOD == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (MOD == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int MOD, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) skel(const float* __restrict__ q, float* __restrict__ pos, float* __restrict__ cost,
                                                   float* __restrict__ gq, long n, int spin, int do_pos) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wblock = (long)blockIdx.x * WAVES + wave, base = wblock * 64;
    if (base >= n) return;
    const f4* q4 = reinterpret_cast<const f4*>(q + base * 7);
    f4 a = q4[lane], b = lane < 48 ? q4[64 + lane] : f4{0, 0, 0, 0};      // 64*7/4 = 112 float4 per wave
    float x = a.x + b.y;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    f4 v = {x, a.y, a.z, b.w};
    if (do_pos) {
        f4* p4 = reinterpret_cast<f4*>(pos + base * 33);                   // 64*33/4 = 528 float4 per wave = 8.25 chunks
#pragma unroll
        for (int j = 0; j < 8; ++j) st<MOD>(p4 + j * 64 + lane, v);
        if (lane < 16) st<MOD>(p4 + 512 + lane, v);
    }
    f4* g4 = reinterpret_cast<f4*>(gq + base * 7);
    st<MOD>(g4 + lane, v);
    if (lane < 48) st<MOD>(g4 + 64 + lane, v);
    cost[base + lane] = x;
}
// BPW consecutive 64-sample blocks per wave (grid shrinks by BPW): does the dispatch ramp shrink with the number of waves?
template <int MOD, int WAVES, int BPW>
__global__ void __launch_bounds__(WAVES * 64) skel_multi(const float* __restrict__ q, float* __restrict__ pos, float* __restrict__ cost,
                                                         float* __restrict__ gq, long n, unsigned long long* stamps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w0 = (long)blockIdx.x * WAVES + wave;
    const long nw = (long)gridDim.x * WAVES;
    if (stamps && lane == 0) stamps[w0 * 2] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const long base = (w0 + i * nw) * 64;
        if (base >= n) break;
        const f4* q4 = reinterpret_cast<const f4*>(q + base * 7);
        f4 a = q4[lane], b = lane < 48 ? q4[64 + lane] : f4{0, 0, 0, 0};
        f4 v = {a.x + b.y, a.y, a.z, b.w};
        f4* p4 = reinterpret_cast<f4*>(pos + base * 33);
#pragma unroll
        for (int j = 0; j < 8; ++j) st<MOD>(p4 + j * 64 + lane, v);
        if (lane < 16) st<MOD>(p4 + 512 + lane, v);
        f4* g4 = reinterpret_cast<f4*>(gq + base * 7);
        st<MOD>(g4 + lane, v);
        if (lane < 48) st<MOD>(g4 + 64 + lane, v);
        cost[base + lane] = v.x;
    }
    if (stamps && lane == 0) stamps[w0 * 2 + 1] = __builtin_amdgcn_s_memrealtime();
}
// Same bytes, but each wave's nine 1 KiB chunks leave `gap` dependent FMAs apart, `per` chunks at a time (the fused kernel's
// trickle): does HBM absorb interleaved 1 KiB pieces as well as 8 KiB runs?
template <int PER>
__global__ void __launch_bounds__(256) skel_trickle(const float* __restrict__ q, float* __restrict__ pos, float* __restrict__ cost,
                                                    float* __restrict__ gq, long n, int gap, int pre) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long base = ((long)blockIdx.x * 4 + wave) * 64;
    if (base >= n) return;
    const f4* q4 = reinterpret_cast<const f4*>(q + base * 7);
    f4 a = q4[lane], b = lane < 48 ? q4[64 + lane] : f4{0, 0, 0, 0};
    float x = a.x + b.y;
    for (int i = 0; i < pre; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    f4* p4 = reinterpret_cast<f4*>(pos + base * 33);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j % PER == 0) for (int i = 0; i < gap * PER; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
        f4 v = {x, a.y, a.z, b.w};
        st<1>(p4 + j * 64 + lane, v);
    }
    f4 v = {x, a.y, a.z, b.w};
    if (lane < 16) st<1>(p4 + 512 + lane, v);
    f4* g4 = reinterpret_cast<f4*>(gq + base * 7);
    st<1>(g4 + lane, v);
    if (lane < 48) st<1>(g4 + 64 + lane, v);
    cost[base + lane] = x;
}
template <int PER> void run_trickle(const float* q, float* pos, float* cost, float* gq, long n, int gap, int pre) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (int)(n / 64 / 4);
    for (int i = 0; i < 200; ++i) skel_trickle<PER><<<grid, 256>>>(q, pos, cost, gq, n, gap, pre);
    hipEventRecord(a);
    for (int i = 0; i < 2000; ++i) skel_trickle<PER><<<grid, 256>>>(q, pos, cost, gq, n, gap, pre);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("trickle: pre %4d FMAs, then %d chunk(s) every %4d FMAs: %6.2f us per launch\n", pre, PER, gap * PER, ms * 1e3f / 2000);
}
template <int BPW> void run_multi(const float* q, float* pos, float* cost, float* gq, long n, unsigned long long* dst) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int waves = (int)(n / 64 / BPW), grid = waves / 4;
    for (int i = 0; i < 200; ++i) skel_multi<1, 4, BPW><<<grid, 256>>>(q, pos, cost, gq, n, nullptr);
    hipEventRecord(a);
    for (int i = 0; i < 2000; ++i) skel_multi<1, 4, BPW><<<grid, 256>>>(q, pos, cost, gq, n, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    skel_multi<1, 4, BPW><<<grid, 256>>>(q, pos, cost, gq, n, dst); hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(waves * 16);
    hipMemcpy(h, dst, waves * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, e1 = 0, x1 = 0;
    for (int w = 0; w < waves; ++w) { if (h[2 * w] < t0) t0 = h[2 * w]; }
    for (int w = 0; w < waves; ++w) { if (h[2 * w] > e1) e1 = h[2 * w]; if (h[2 * w + 1] > x1) x1 = h[2 * w + 1]; }
    printf("blocks/wave %d (%d waves): %6.2f us per launch | last wave entry +%.2f us, last exit +%.2f us after the first entry\n",
           BPW, waves, ms * 1e3f / 2000, (e1 - t0) / 100.0, (x1 - t0) / 100.0);
    free(h);
}
template <int MOD, int WAVES> float run(const float* q, float* pos, float* cost, float* gq, long n, int spin, int do_pos, int steps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (int)((n / 64 + WAVES - 1) / WAVES);
    for (int i = 0; i < 200; ++i) skel<MOD, WAVES><<<grid, WAVES * 64>>>(q, pos, cost, gq, n, spin, do_pos);
    hipEventRecord(a);
    for (int i = 0; i < steps; ++i) skel<MOD, WAVES><<<grid, WAVES * 64>>>(q, pos, cost, gq, n, spin, do_pos);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / steps;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 4096L * 64;
    float *q, *pos, *cost, *gq;
    hipMalloc(&q, n * 28); hipMalloc(&pos, n * 132); hipMalloc(&cost, n * 4); hipMalloc(&gq, n * 28);
    hipMemset(q, 0, n * 28);
    const int steps = 2000;
    const double mb = n * 192 / 1e6;
    printf("n = %ld samples, %.1f MB per launch\n", n, mb);
    const char* names[4] = {"plain", "sc1", "nt", "sc0 sc1"};
    float t;
#define R(MOD, W, SPIN, POS) t = run<MOD, W>(q, pos, cost, gq, n, SPIN, POS, steps); \
    printf("%-8s waves/wg %d spin %5d pos %d : %6.2f us  (%.0f GB/s of the %s bytes)\n", names[MOD], W, SPIN, POS, t, \
           (POS ? mb : n * 60 / 1e6) / t * 1e3, POS ? "192 B/sample" : "60 B/sample");
    R(0, 4, 0, 1) R(1, 4, 0, 1) R(2, 4, 0, 1) R(3, 4, 0, 1)
    R(1, 1, 0, 1) R(1, 2, 0, 1) R(1, 8, 0, 1) R(1, 16, 0, 1)
    R(1, 4, 0, 0) R(0, 4, 0, 0)
    // dependent FMA chain: ~10 ns each at 4 waves per SIMD (see spin rows above) -> pre 150 ~ 1.5 us, gap 30 ~ 0.3 us
    for (int pre : {0, 150}) for (int gap : {0, 15, 30, 60}) { run_trickle<1>(q, pos, cost, gq, n, gap, pre); }
    for (int gap : {15, 30, 60}) { run_trickle<2>(q, pos, cost, gq, n, gap, 150); run_trickle<4>(q, pos, cost, gq, n, gap, 150); }
    // no stores at all would take: pre + 8 * gap FMAs
    unsigned long long* stamps; hipMalloc(&stamps, 4096 * 16);
    run_multi<1>(q, pos, cost, gq, n, stamps); run_multi<2>(q, pos, cost, gq, n, stamps); run_multi<4>(q, pos, cost, gq, n, stamps);
    run_multi<8>(q, pos, cost, gq, n, stamps);
    return 0;
}
