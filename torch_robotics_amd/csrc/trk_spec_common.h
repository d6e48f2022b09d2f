// trk_spec_common.h -- pieces shared by the model-specialised (generated) fused rollout kernels.
//
// A generated kernel (torch_robotics_amd/codegen.py) is straight-line code for ONE kinematic tree and ONE
// set of collision links: the chain is unrolled, URDF constants are literals, structural zeros and +-1 are
// folded away, every pose lives in named registers.  What stays runtime (wave-uniform, scalar-loaded) is the
// scene: primitives, margins, workspace box, EE target, weights, base pose.
//
// Work mapping: one 64-lane wavefront = 64 consecutive samples (one trajectory at horizon 64); SPEC_WAVES
// wavefronts share a workgroup only to cut the number of workgroups the dispatcher has to place (they never
// exchange data: each wave owns a private LDS region, `__syncthreads()` is just the cheapest fence).
// LDS (8.25 KiB per wave for Panda) is used only to transpose between "lane owns a sample" and "wave writes a
// contiguous span": q in, link positions / gradient out -- every HBM access of a wave is a contiguous run.
#define SPEC_WAVES 4
#define SPEC_BLOCK (SPEC_WAVES * TRK_WAVE)
#pragma once
#include "trk_device.h"

struct SpecArgs {
    DevCostHdr C;
    TrkRolloutWeights w;
    float base_R[9];
    float base_t[3];
    const void* q;                // float or _Float16 (io_f16), like link_pos and gq
    int64_t n;
    void* link_pos;
    float* cost;
    void* gq;
    float* cost_sum;
    int32_t io_f16;               // 1: q / link_pos / gq are fp16 in HBM (arithmetic, cost and cost_sum stay fp32)
    unsigned long long* stamps;   // profiling hook (nullable): [n_waves][8] s_memtime stamps at phase boundaries
};

typedef void (*SpecLaunchFn)(const SpecArgs& args, int base_identity, hipStream_t stream);

// Layout version of SpecArgs / SpecEntry / DevCostHdr as seen by a generated unit.  A unit compiled against another layout
// (a stale on-disk JIT object) must never be dispatched: trk_spec_register refuses it.  Bump on ANY change to these structs,
// to TrkRolloutWeights or to the TRK_MAX_* limits in include/trk.h.
#define TRK_SPEC_ABI_VERSION (TRK_ABI_VERSION * 1000 + 2)

struct SpecEntry {
    int32_t spec_abi_version;   // TRK_SPEC_ABI_VERSION the unit was compiled with
    uint32_t sizeof_args;       // sizeof(SpecArgs) the unit was compiled with
    uint32_t sizeof_entry;      // sizeof(SpecEntry) the unit was compiled with
    uint32_t sizeof_cost_hdr;   // sizeof(DevCostHdr)
    uint64_t model_hash;        // FNV-1a over the kinematic tables (see trk_capi.hip: model_hash)
    int32_t n_links, n_dofs;
    int32_t n_obj_links;        // baked collision-link template
    const int32_t* obj_link_idx;   // LINK indices -- COLUMN indices when n_points > 0
    int32_t n_self_pairs;
    const int32_t* self_pairs;  // [2*P] LINK indices (already mapped through self_link_idx) -- COLUMNS when n_points > 0
    int32_t ee_link;
    const char* name;
    SpecLaunchFn launch;
    int32_t n_points;           // 0: the kernel's columns are the link origins; else: a baked attached-point set
    uint64_t points_hash;       // FNV-1a over (n_points, point_link[], point_offset[]) in the caller's order
    SpecLaunchFn launch_posbwd; // reverse mode of the link positions (q, gpos = link_pos -> gq); nullptr if not generated
    int32_t ee2_link;           // second tracked link baked into the unit (-1 = none)
};

// registry filled by static initialisers of the generated translation units
// returns 0 when the unit was accepted, TRK_ERR_INVALID_ARG (and registers nothing) when its layout stamp differs
int trk_spec_register(const SpecEntry* e);
#define SPEC_ENTRY_STAMP TRK_SPEC_ABI_VERSION, (uint32_t)sizeof(SpecArgs), (uint32_t)sizeof(SpecEntry), (uint32_t)sizeof(DevCostHdr)
const SpecEntry* trk_spec_find(uint64_t model_hash, int n_links, int n_dofs);
const SpecEntry* trk_spec_find_points(uint64_t model_hash, uint64_t points_hash, int n_points);

// ---------------------------------------------------------------------------------------------------------
// I/O transposes (one wavefront, `lane` = lane id; `lds` = this wave's private region)
// ---------------------------------------------------------------------------------------------------------
// Ordering between this wave's own LDS writes and reads: DS instructions of one wave execute in order, so no
// s_barrier is needed -- only a compiler-level fence (a workgroup barrier would make the four waves of a
// workgroup march in lockstep and serialise their memory phases).
__device__ __forceinline__ void spec_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Output stores are write-through (sc1): a plain store leaves its line dirty in the XCD's L2, and at the kernel
// boundary up to 32 MB of dirty lines must be written back before the next step's loads get through -- measured as a
// 2.7 us stall of every wave's first q load.  Write-through lets the 43 MB drain while the kernel computes.
typedef float trk_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt_f4(float4* p, const float4& v) {
    const trk_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(x) : "memory");
}
typedef float trk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_wt_f2(float* p, float a, float b) {
    const trk_f2 x = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void store_wt_f1(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(v) : "memory");
}

// HBM-side element type of q / link_pos / gq: four consecutive elements <-> one float4 of LDS (fp32 arithmetic always).
typedef _Float16 trk_h4 __attribute__((ext_vector_type(4)));
template <class IO> struct IoQuad;
template <> struct IoQuad<float> {
    static constexpr uintptr_t kAlignMask = 15;
    static __device__ __forceinline__ float4 load(const float* p, int k) { return reinterpret_cast<const float4*>(p)[k]; }
    static __device__ __forceinline__ void store_wt(float* p, int k, const float4& v) { store_wt_f4(reinterpret_cast<float4*>(p) + k, v); }
    static __device__ __forceinline__ void store_wt1(float* p, float v) { store_wt_f1(p, v); }
    static __device__ __forceinline__ void store_wt2(float* p, float a, float b) { store_wt_f2(p, a, b); }
};
template <> struct IoQuad<_Float16> {
    static constexpr uintptr_t kAlignMask = 7;
    static __device__ __forceinline__ float4 load(const _Float16* p, int k) {
        const trk_h4 h = reinterpret_cast<const trk_h4*>(p)[k];
        return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
    }
    static __device__ __forceinline__ void store_wt(_Float16* p, int k, const float4& v) {
        const trk_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(reinterpret_cast<trk_h4*>(p) + k), "v"(h) : "memory");
    }
    static __device__ __forceinline__ void store_wt1(_Float16* p, float v) {
        const _Float16 h = (_Float16)v;
        asm volatile("global_store_short %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(h) : "memory");
    }
    static __device__ __forceinline__ void store_wt2(_Float16* p, float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = {(_Float16)a, (_Float16)b};
        asm volatile("global_store_dword %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(h) : "memory");
    }
};

// copy the first TRK_LDS_SPHERES world-frame spheres into this wave's LDS (one 16-byte load per lane, issued together
// with the q loads so its latency is free)
__device__ __forceinline__ void spec_load_spheres(const DevCostHdr& C, float4* lds_spheres, int lane) {
    if (lane < TRK_LDS_SPHERES && lane < C.n_spheres) lds_spheres[lane] = C.spheres[lane];
}

template <int D, class IO>
__device__ __forceinline__ void spec_load_q(const IO* __restrict__ q, int64_t base, int rows, int lane,
                                            float* lds, float (&qv)[D]) {
    // the wave's 64*D floats are one contiguous span: 16-byte loads (2 instructions for D = 7) into LDS, then a
    // stride-D read back (D odd -> conflict-free); ragged / unaligned tails take the dword path
    const int64_t first = base * D;
    const IO* src = q + first;
    constexpr int NV = TRK_WAVE * D / 4;
    if (rows == TRK_WAVE && (TRK_WAVE * D) % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) & IoQuad<IO>::kAlignMask) == 0)) {
        float4* lds4 = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int j = 0; j < (NV + TRK_WAVE - 1) / TRK_WAVE; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < NV) lds4[k] = IoQuad<IO>::load(src, k);
        }
    } else {
        const int count = rows * D;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = lane + TRK_WAVE * j;
            lds[k] = k < count ? (float)src[k] : 0.0f;
        }
    }
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) qv[j] = lds[lane * D + j];
    spec_wave_sync();
}

template <int D, class IO>
__device__ __forceinline__ void spec_store_gq(IO* __restrict__ gq, int64_t base, int rows, int lane,
                                              float* lds, const float (&gv)[D]) {
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) lds[lane * D + j] = gv[j];
    spec_wave_sync();
    const int64_t first = base * D;
    IO* dst = gq + first;
    constexpr int NV = TRK_WAVE * D / 4;
    if (rows == TRK_WAVE && (TRK_WAVE * D) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0)) {
        const float4* lds4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
        for (int j = 0; j < (NV + TRK_WAVE - 1) / TRK_WAVE; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < NV) IoQuad<IO>::store_wt(dst, k, lds4[k]);
        }
    } else {
        const int count = rows * D;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < count) IoQuad<IO>::store_wt1(dst + k, lds[k]);
        }
    }
}

// Link positions: each lane writes its 3L floats at stride 3L (33 for Panda: conflict-free), then the wave
// streams the 64*3L contiguous floats out in 1 KiB chunks (one ds_read_b128 + one global_store_dwordx4 per
// lane and chunk).  The chunks are NOT issued back to back: every wave of the chip reaches this point at the same
// time, and 16 waves x 8.25 KiB per CU saturate the store path for ~6 us during which nothing computes.
// `tick()` issues one chunk; the kernel calls it between blocks of arithmetic so the 34.6 MB trickle out at
// roughly the rate HBM absorbs them.
template <int W, class IO>
struct PosFlusher {
    static constexpr int NV = W * TRK_WAVE / 4;                  // 4-element chunks per wave
    static constexpr int NCHUNK = (NV + TRK_WAVE - 1) / TRK_WAVE;
    const float4* src4;
    IO* dst;
    int lane;
    int next;                                                    // wave-uniform
    __device__ __forceinline__ void operator()() {
        if (next < NCHUNK) {
            const int k = lane + TRK_WAVE * next;
            if (k < NV) IoQuad<IO>::store_wt(dst, k, src4[k]);
            ++next;
        }
    }
    __device__ __forceinline__ void flush() {
        while (next < NCHUNK) (*this)();
    }
};

// stand-in for PosFlusher in kernels whose positions leave through spec_flush_chunk instead
struct NoFlush {
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void flush() const {}
};

// stage the wave's rows in LDS; returns a flusher (fast path) or writes everything now (ragged / unaligned tail)
template <int W, class IO>
__device__ __forceinline__ PosFlusher<W, IO> spec_stage_rows(IO* __restrict__ out, int64_t base, int rows, int lane,
                                                             float* lds, const float (&v)[W]) {
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < W; ++j) lds[lane * W + j] = v[j];
    spec_wave_sync();
    IO* dst = out + base * W;
    PosFlusher<W, IO> f{reinterpret_cast<const float4*>(lds), dst, lane, 0};
    const bool fast = rows == TRK_WAVE && (W * TRK_WAVE) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0);
    if (!fast) {
        const int count = rows * W;
        for (int k = lane; k < count; k += TRK_WAVE) dst[k] = (IO)lds[k];
        f.next = PosFlusher<W, IO>::NCHUNK;
    }
    return f;
}

// Wide rows (attached points: W = 3P floats per sample) do not fit a whole-row staging buffer, so they leave in column
// chunks: a chunk is the NF consecutive floats [c0, c0 + NF) of every sample's row.  Each lane has put its NF floats at
// lds[lane * LS ...]; the wave then streams the rows' segments as V-float vectors (V | NF, V | W, V | c0: 16/8/4-byte
// aligned), vector e -> sample e / (NF / V).  A sample's segment is contiguous, neighbouring chunks complete its lines.
template <int W, int NF, int LS, int V, class IO = float>
__device__ __forceinline__ void spec_flush_chunk(IO* __restrict__ out, int64_t base, int c0, int rows, int lane,
                                                 const float* lds) {
    constexpr int NVEC = NF / V;
    static_assert(NF % V == 0 && W % V == 0 && LS % V == 0, "chunk geometry must keep the vectors aligned");
    spec_wave_sync();
    const int total = rows * NVEC;
    IO* dst0 = out + base * W + c0;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
        const int e = lane + TRK_WAVE * j;
        if (e < total) {
            const int smp = e / NVEC, v = e - smp * NVEC;
            const float* src = lds + smp * LS + v * V;
            IO* dst = dst0 + (int64_t)smp * W + v * V;
            if (V == 4) IoQuad<IO>::store_wt(dst, 0, *reinterpret_cast<const float4*>(src));
            else if (V == 2) IoQuad<IO>::store_wt2(dst, src[0], src[1]);
            else IoQuad<IO>::store_wt1(dst, src[0]);
        }
    }
    spec_wave_sync();           // the chunk buffer may be overwritten from here on
}

// mirror of spec_flush_chunk: floats [c0, c0 + NF) of the wave's rows -> lds[smp * LS + ...]; each lane then reads its own row
template <int W, int NF, int LS, int V>
__device__ __forceinline__ void spec_load_chunk(const float* __restrict__ in, int64_t base, int c0, int rows, int lane,
                                                float* lds) {
    constexpr int NVEC = NF / V;
    static_assert(NF % V == 0 && W % V == 0 && LS % V == 0, "chunk geometry must keep the vectors aligned");
    spec_wave_sync();           // everybody has consumed the previous chunk
    const int total = rows * NVEC;
    const float* src0 = in + base * W + c0;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
        const int e = lane + TRK_WAVE * j;
        if (e < total) {
            const int smp = e / NVEC, v = e - smp * NVEC;
            const float* src = src0 + (int64_t)smp * W + v * V;
            float* dst = lds + smp * LS + v * V;
            if (V == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
            else if (V == 2) *reinterpret_cast<float2*>(dst) = *reinterpret_cast<const float2*>(src);
            else dst[0] = src[0];
        }
    }
    spec_wave_sync();
}

// same, for rows the kernel has already written to lds[lane * W + j] piecemeal (robots with many links: staging each
// link's position as soon as it exists keeps 3L values from being live at once)
template <int W, class IO>
__device__ __forceinline__ PosFlusher<W, IO> spec_stage_rows_prefilled(IO* __restrict__ out, int64_t base, int rows, int lane,
                                                                       float* lds) {
    spec_wave_sync();
    IO* dst = out + base * W;
    PosFlusher<W, IO> f{reinterpret_cast<const float4*>(lds), dst, lane, 0};
    const bool fast = rows == TRK_WAVE && (W * TRK_WAVE) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0);
    if (!fast) {
        const int count = rows * W;
        for (int k = lane; k < count; k += TRK_WAVE) dst[k] = (IO)lds[k];
        f.next = PosFlusher<W, IO>::NCHUNK;
    }
    return f;
}

// profiling hook: lane 0 of a wave records the shader clock at phase `k` (no-op when A.stamps == nullptr)
__device__ __forceinline__ void spec_stamp(unsigned long long* stamps, int64_t wblock, int k, int lane) {
    if (stamps) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamps[wblock * 8 + k] = t;
    }
}

__device__ __forceinline__ float spec_wave_sum(float v) { return trk_wave_sum(v); }

// ---------------------------------------------------------------------------------------------------------
// collision objectives on NL link points held in registers.  Adds w * cost to `cost` and
// w * d cost / d p to (gx, gy, gz) (accumulating).  Margins are C.obj_link_margin[0..NL) in baked order.
// ---------------------------------------------------------------------------------------------------------
template <int NL, class Tick, bool FAST = false>
__device__ __forceinline__ float spec_objects_cost(const DevCostHdr& C, float w, const float (&px)[NL],
                                                   const float (&py)[NL], const float (&pz)[NL], float (&gx)[NL],
                                                   float (&gy)[NL], float (&gz)[NL], Tick& tick, const float4* lds_spheres,
                                                   int mbase = 0) {
    float s[NL], ax[NL], ay[NL], az[NL];
    scene_min_sdf<NL, Tick&, FAST>(C, px, py, pz, s, ax, ay, az, tick, lds_spheres);
    float cost = 0.0f;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        cost += cptr(C.obj_link_margin)[mbase + l] - s[l];                     // sum_l max_o (margin - sdf_o) = sum_l (margin - min_o sdf_o)
        gx[l] = fmaf(-w, ax[l], gx[l]); gy[l] = fmaf(-w, ay[l], gy[l]); gz[l] = fmaf(-w, az[l], gz[l]);
    }
    return w * cost;
}

template <int NL>
__device__ __forceinline__ float spec_ws_cost(const DevCostHdr& C, float w, const float (&px)[NL], const float (&py)[NL],
                                              const float (&pz)[NL], float (&gx)[NL], float (&gy)[NL], float (&gz)[NL],
                                              int mbase = 0) {
    float cost = 0.0f;
#pragma unroll
    for (int l = 0; l < NL; ++l) cost += ws_cost_point(C, cptr(C.obj_link_margin)[mbase + l], px[l], py[l], pz[l], w, gx[l], gy[l], gz[l]);
    return w * cost;
}

// one self-collision pair (distance_fields.py:194-208): returns w*(margin - ||pa - pb||), accumulates gradients
__device__ __forceinline__ float spec_self_pair(float w, float margin, float ax, float ay, float az, float bx, float by,
                                                float bz, float& gax, float& gay, float& gaz, float& gbx, float& gby,
                                                float& gbz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rs = n2 > 0.0f ? trk_rsq(n2) : 0.0f;              // one transcendental: 1/||d|| (0 at d = 0, like torch.norm's backward)
    const float nrm = n2 * rs;
    const float inv = w * rs;
    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
    gax -= ux; gay -= uy; gaz -= uz;
    gbx += ux; gby += uy; gbz += uz;
    return w * (margin - nrm);
}
