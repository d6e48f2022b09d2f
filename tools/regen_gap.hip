// How long is a wave slot empty between a workgroup's exit and the entry of the workgroup that replaces it?
// 1024 workgroups x 256 threads, dynamic LDS sized so that TWO fit a CU (512 resident, two generations).  Each wave spins for
// `spin` microseconds, then writes `nstore` x 1 KiB (one store instruction each) with the chosen store flavour and exits.
// Per wave: entry / exit by s_memrealtime (100 MHz, chip-wide) + HW_ID.  Host: for every (CU, SIMD, wave slot) the gap
// between the first generation's exit and the second generation's entry.
// build: hipcc -O3 --offload-arch=gfx950 tools/regen_gap.hip -o tools/regen_gap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
struct Pad { float v[120]; };          // by-value kernel arguments: 480 bytes, about the size of SpecArgs
template <int MODE, bool PAD>          // 0: no stores, 1: plain stores, 2: sc1 write-through, 3: nt
__global__ void __launch_bounds__(256) k(unsigned long long* stamps, float4* out, Pad pad, int spin_ticks, int nstore) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_ID
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;  // XCC_ID
    float x = threadIdx.x;
    if (PAD) x += pad.v[0] + pad.v[40] + pad.v[80] + pad.v[119];
    lds[threadIdx.x] = x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) { x = __builtin_fmaf(x, 1.0000001f, 1e-9f); __builtin_amdgcn_s_sleep(8); }
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {x, x + 1, x + 2, x + 3};
    float4* dst = out + (w * nstore) * 64 + lane;
    for (int i = 0; i < nstore; ++i) {
        if (MODE == 1) dst[i * 64] = make_float4(v.x, v.y, v.z, v.w);
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i * 64), "v"(v));
        if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst + i * 64), "v"(v));
    }
    if (lane == 0) {
        stamps[4 * w] = t0; stamps[4 * w + 1] = __builtin_amdgcn_s_memrealtime(); stamps[4 * w + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}
template <int MODE, bool PAD>
void run(const char* name, unsigned long long* d, float4* out, int spin_us, int nstore, int lds_bytes) {
    Pad pad{};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) k<MODE, PAD><<<1024, 256, lds_bytes>>>(d, out, pad, spin_us * 100, nstore);
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) k<MODE, PAD><<<1024, 256, lds_bytes>>>(d, out, pad, spin_us * 100, nstore);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(4096 * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> slots;
    unsigned long long t0 = ~0ull, tl = 0;
    for (int w = 0; w < 4096; ++w) {
        const unsigned long long id = h[4 * w + 2] & 0xf0000ffffull;       // xcc | se, sh, cu, simd, wave slot
        slots[id].push_back({h[4 * w], h[4 * w + 1]});
        t0 = std::min(t0, h[4 * w]); tl = std::max(tl, h[4 * w + 1]);
    }
    std::vector<double> gaps;
    for (auto& kv : slots) {
        auto& v = kv.second; std::sort(v.begin(), v.end());
        for (size_t i = 1; i < v.size(); ++i) gaps.push_back(((double)v[i].first - (double)v[i - 1].second) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    double mean = 0; for (double g : gaps) mean += g; mean /= gaps.empty() ? 1 : gaps.size();
    printf("%-52s %7.2f us/launch | span %6.2f us | slots %4zu | regen gap mean %5.2f p5 %5.2f p50 %5.2f p95 %5.2f us (n=%zu)\n", name, ms * 5.0f,
           (tl - t0) / 100.0, slots.size(), mean, gaps.empty() ? 0 : gaps[gaps.size() / 20], gaps.empty() ? 0 : gaps[gaps.size() / 2],
           gaps.empty() ? 0 : gaps[gaps.size() * 19 / 20], gaps.size());
}
int main() {
    unsigned long long* d; float4* out;
    hipMalloc(&d, 4096 * 32); hipMalloc(&out, (size_t)4096 * 64 * 64 * 16);
    const int lds = 72 * 1024;          // two workgroups per CU
    run<0, false>("no stores, small kernarg", d, out, 5, 0, lds);
    run<0, true>("no stores, 480 B kernarg (read)", d, out, 5, 0, lds);
    run<1, true>("4 plain stores at the end", d, out, 5, 4, lds);
    run<2, true>("4 sc1 stores at the end", d, out, 5, 4, lds);
    run<3, true>("4 nt stores at the end", d, out, 5, 4, lds);
    run<1, true>("32 plain stores at the end (134 MB / launch)", d, out, 5, 32, lds);
    run<2, true>("32 sc1 stores at the end", d, out, 5, 32, lds);
    run<3, true>("32 nt stores at the end", d, out, 5, 32, lds);
    run<0, true>("no stores, 8 KB LDS (all 4096 waves resident)", d, out, 5, 0, 8 * 1024);
    return 0;
}
