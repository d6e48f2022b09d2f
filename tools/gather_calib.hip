// gather_calib.hip -- calibration of the HBM counters (FETCH_SIZE and its raw parts) and of the DRAM's gather rate for the access
// pattern of the voxel-grid scene: one 16-byte (or 8-byte) record per lane at an independent random position of a table far larger
// than the 256 MB Infinity Cache.  VERDICT r4: the x 1.999 FETCH correction was calibrated on a streaming read only.
//
//   gather_calib.bin [table_MiB=4096] [gathers_per_lane=8] [reps=20]
//
// Per launch: n_lanes x gathers_per_lane gathers of REC bytes (algorithmic bytes = that x REC) + 4 bytes written per lane.
// Every gather lands in a different 128-byte line (index = odd-multiplier hash of the gather number, times 8 records), so no two
// gathers of a launch share a line or a sector: whatever the counters report per gather IS the request granularity as counted, and
// gathers / second x (true bytes per request) cannot exceed what the DRAM delivers -- the rate bounds the true request size.
// Prints one line per variant: kernel time (HIP events), gathers / s, algorithmic GB/s.  Run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE            (and, separate passes:)
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
// to get the counters per launch of k_gather<16> / k_gather<8> / k_stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int REC> struct RecT;
template <> struct RecT<16> { typedef float4 T; static __device__ float sum(const T& v) { return (v.x + v.y) + (v.z + v.w); } };
template <> struct RecT<8> { typedef float2 T; static __device__ float sum(const T& v) { return v.x + v.y; } };

// lines: number of 128-byte lines of the table (a power of two); gather g reads the first REC bytes of line hash(g)
template <int REC, int K>
__global__ void __launch_bounds__(256) k_gather(const char* __restrict__ table, unsigned line_mask, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    typedef typename RecT<REC>::T T;
    T v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned g = gid * (unsigned)K + (unsigned)k;
        const unsigned line = (g * 2654435761u) & line_mask;            // odd multiplier: a bijection on 2^32, distinct lines per launch
        v[k] = *reinterpret_cast<const T*>(table + (size_t)line * 128u);
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) acc += RecT<REC>::sum(v[k]);
    out[gid] = acc;
}

// the streaming reference: the same bytes per lane as 16-byte coalesced loads
template <int K>
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ table, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) { const float4 v = table[(size_t)k * gridDim.x * 256u + gid]; acc += (v.x + v.y) + (v.z + v.w); }
    out[gid] = acc;
}

template <class F>
static int timed(const char* name, int reps, double gathers, double bytes, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double s = ms * 1e-3 / reps;
    printf("%-14s %9.1f us/launch  %8.2f G gathers/s  %8.1f GB/s algorithmic  (a 32 / 64 / 128-byte request per gather would be %.0f / %.0f / %.0f GB/s)\n",
           name, s * 1e6, gathers / s / 1e9, bytes / s / 1e9, gathers * 32 / s / 1e9, gathers * 64 / s / 1e9, gathers * 128 / s / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 4096;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    size_t lines = 1;
    while (lines * 2 * 128 <= mib * 1024 * 1024) lines *= 2;
    const size_t bytes = lines * 128;
    constexpr int K = 8;
    const unsigned n_lanes = 8u * 1024u * 1024u;                        // 64 M gathers per launch
    char* table = nullptr; float* out = nullptr;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, (size_t)n_lanes * 4));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipDeviceSynchronize());
    printf("table %zu MiB (%zu lines of 128 B), %u lanes x %d gathers per launch, %d launches timed\n", bytes >> 20, lines, n_lanes, K, reps);
    const double G = (double)n_lanes * K;
    const unsigned mask = (unsigned)(lines - 1);
    const dim3 grid(n_lanes / 256), block(256);
    if (timed("gather 16 B", reps, G, G * 16, [&] { hipLaunchKernelGGL((k_gather<16, K>), grid, block, 0, 0, table, mask, out); })) return 1;
    if (timed("gather 8 B", reps, G, G * 8, [&] { hipLaunchKernelGGL((k_gather<8, K>), grid, block, 0, 0, table, mask, out); })) return 1;
    if ((size_t)n_lanes * K * 16 <= bytes)
        if (timed("stream 16 B", reps, G, G * 16, [&] { hipLaunchKernelGGL((k_stream<K>), grid, block, 0, 0, reinterpret_cast<const float4*>(table), out); })) return 1;
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
