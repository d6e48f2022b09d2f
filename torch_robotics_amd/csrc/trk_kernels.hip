// trk_kernels.hip -- table-driven gfx950 kernels: any URDF tree, any cost model.
//
// Mapping: one lane per (batch, horizon) sample, one 64-lane wavefront per workgroup
// (a wavefront is one trajectory when horizon == 64).  The model tables are wave-uniform and
// arrive through scalar loads; per-lane state lives in VGPRs; LDS holds the q / gq / position
// tiles (so every HBM access of a wavefront is one contiguous span) plus the branch-pose stack
// and the per-joint records of the reverse pass.
//
// Reverse mode uses the transpose of the geometric Jacobian instead of matrix adjoints: every
// link contributes a wrench w_i = (tbar_i, t_i x tbar_i + axial(Rbar_i R_i^T)); a revolute joint's
// gradient is  s * z_j . (tau_sub - t_j x f_sub)  over the wrenches of its subtree, a prismatic
// joint's is  (R_parent axis) . f_sub.  In DFS pre-order a subtree is a contiguous range, so the
// subtree sum is a difference of two running prefix sums and one forward walk suffices
// (equals autograd through rigid_body.py:146-211).
#include <algorithm>
#include "trk_device.h"

// k / w for 0 <= k < 2^22 and a wave-uniform w: one multiply by the reciprocal and a fix-up instead of a 32-bit division
__device__ __forceinline__ int trk_div_small(int k, int w, float inv_w) {
    int s = (int)((float)k * inv_w);
    s -= (s * w > k); s += ((s + 1) * w <= k);
    return s;
}

namespace {

// IO = HBM-side element type (float, or _Float16 for the fp16-I/O rollout); LDS and arithmetic are fp32.
// Loads are issued in batches of TRK_LOAD_BATCH before the first one is consumed: a plain `dst[k] = src[k]` loop
// compiles to load / s_waitcnt vmcnt(0) / ds_write per trip, i.e. ONE 256-byte request in flight per wavefront --
// measured 30 us for the 35 MB of k_cost_fields (33 serialized round trips per wavefront).
#define TRK_LOAD_BATCH 8
template <class IO>
__device__ __forceinline__ void load_tile(float* dst, const IO* __restrict__ src, int64_t first, int64_t count, int lane) {
    // dst[0..count) = src[first .. first+count), contiguous -> fully coalesced loads
    for (int64_t k0 = lane; k0 < count; k0 += TRK_WAVE * TRK_LOAD_BATCH) {
        float v[TRK_LOAD_BATCH];
#pragma unroll
        for (int j = 0; j < TRK_LOAD_BATCH; ++j) {
            const int64_t k = k0 + TRK_WAVE * j;
            v[j] = k < count ? (float)src[first + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TRK_LOAD_BATCH; ++j) {
            const int64_t k = k0 + TRK_WAVE * j;
            if (k < count) dst[k] = v[j];
        }
    }
}

template <class IO>
__device__ __forceinline__ void store_tile(IO* __restrict__ dst, const float* src, int64_t first, int64_t count, int lane) {
    for (int64_t k = lane; k < count; k += TRK_WAVE) dst[first + k] = (IO)src[k];
}

// gradient rows of the fp16-q rollout: multiplied by the caller's grad_scale, an fp16 element saturates at +-65504 (no inf)
__device__ __forceinline__ void put_scaled(float* p, float v) { *p = v; }
__device__ __forceinline__ void put_scaled(_Float16* p, float v) { *p = (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
template <class G>
__device__ __forceinline__ void store_tile_scaled(G* __restrict__ dst, const float* src, int64_t first, int64_t count, int lane, float scale) {
    for (int64_t k = lane; k < count; k += TRK_WAVE) put_scaled(dst + first + k, src[k] * scale);
}

// copy a [rows][width] tile kept in LDS with an odd row stride `rs` to/from contiguous global memory
template <class IO>
__device__ __forceinline__ void store_tile_strided(IO* __restrict__ dst, const float* src, int64_t first, int rows,
                                                   int width, int rs, int lane) {
    int r = lane / width, c = lane - r * width;
    const int dr = TRK_WAVE / width, dc = TRK_WAVE - dr * width;
    const int64_t count = (int64_t)rows * width;
    for (int64_t k = lane; k < count; k += TRK_WAVE) {
        dst[first + k] = (IO)src[r * rs + c];
        r += dr; c += dc;
        if (c >= width) { c -= width; ++r; }
    }
}
__device__ __forceinline__ void load_tile_strided(float* dst, const float* __restrict__ src, int64_t first, int rows,
                                                  int width, int rs, int lane) {
    int r = lane / width, c = lane - r * width;
    const int dr = TRK_WAVE / width, dc = TRK_WAVE - dr * width;
    const int64_t count = (int64_t)rows * width;
    for (int64_t k0 = lane; k0 < count; k0 += TRK_WAVE * TRK_LOAD_BATCH) {
        float v[TRK_LOAD_BATCH];
#pragma unroll
        for (int j = 0; j < TRK_LOAD_BATCH; ++j) {
            const int64_t k = k0 + TRK_WAVE * j;
            v[j] = k < count ? src[first + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < TRK_LOAD_BATCH; ++j) {
            const int64_t k = k0 + TRK_WAVE * j;
            if (k < count) dst[r * rs + c] = v[j];
            r += dr; c += dc;
            if (c >= width) { c -= width; ++r; }
        }
    }
}

// One link record through the scalar cache (2 x s_load_dwordx16).  The walks fetch the record of position p + 1
// while they work on position p: a record loaded at its first use stalls the wavefront for a scalar-cache round trip
// per link, and nothing else hides it when only 2-3 wavefronts share a SIMD.
__device__ __forceinline__ DevLink load_link(const DevLink* __restrict__ links, int p) {
    const TRK_CAS DevLink* c = cptr(links) + p;
    DevLink L;
#pragma unroll
    for (int k = 0; k < 9; ++k) L.Rf[k] = c->Rf[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { L.trans[k] = c->trans[k]; L.axis[k] = c->axis[k]; }
    L.lower = c->lower; L.upper = c->upper; L.rot_sign = c->rot_sign;
    L.type = c->type; L.dof = c->dof; L.rot_axis = c->rot_axis; L.clamp = c->clamp;
    L.parent_slot = c->parent_slot; L.store_slot = c->store_slot; L.link = c->link;
    L.sf_rot_axis = c->sf_rot_axis; L.sf_clamp = c->sf_clamp; L.jac_axis = c->jac_axis;
    L.fin_begin = c->fin_begin; L.fin_end = c->fin_end;
    return L;
}

__device__ __forceinline__ void slot_store(float* slots, int slot, int lane, const Pose& p) {
    float* s = slots + slot * 12 * TRK_WAVE + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k * TRK_WAVE] = p.r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[(9 + k) * TRK_WAVE] = p.t[k];
}
__device__ __forceinline__ void slot_load(const float* slots, int slot, int lane, Pose& p) {
    const float* s = slots + slot * 12 * TRK_WAVE + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) p.r[k] = s[k * TRK_WAVE];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.t[k] = s[(9 + k) * TRK_WAVE];
}

__device__ __forceinline__ float wave_sum(float v) { return trk_wave_sum(v); }

// One step of the forward walk: pose of the link at pre-order position p.
template <bool STATEFUL>
__device__ __forceinline__ float walk_step(const DevModelHdr& hdr, const DevLink& L, int p, const float* qs, int D,
                                           int lane, float* slots, Pose& cur, Pose& par_out) {
    float pass = 1.0f;
    if (p == 0) {
        pose_from_base(hdr, cur);
        par_out = cur;
    } else {
        Pose par;
        if (L.parent_slot >= 0) slot_load(slots, L.parent_slot, lane, par);
        else par = cur;
        const float q = L.dof >= 0 ? qs[lane * D + L.dof] : 0.0f;
        pass = joint_compose<STATEFUL>(L, par, q, cur);
        par_out = par;
    }
    if (L.store_slot >= 0) slot_store(slots, L.store_slot, lane, cur);
    return pass;
}

}  // namespace

// ============================================================================================
// FK forward.  MODE 0: H [N, n_sel, 4, 4] (robot_tree.py:267-301); MODE 1: positions [N, n_sel, 3]
// (robot_panda.py:138-170).  LDS: q tile | pose slots | (MODE 1) output tile.
// ============================================================================================
template <int MODE>
__global__ void __launch_bounds__(TRK_WAVE)
k_fk_forward(DevModelHdr hdr, const DevLink* __restrict__ links, SelMap sel, int n_sel,
             const float* __restrict__ q, int64_t n, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs, L = hdr.n_links;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* qs = smem;
    float* slots = qs + TRK_WAVE * D;
    float* tile = slots + hdr.n_slots * 12 * TRK_WAVE;
    const int width = n_sel * 3, rs = width | 1;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    __syncthreads();
    const int64_t s = base + lane;
    Pose cur, par;
    DevLink nxt = load_link(links, 0);
    for (int p = 0; p < L; ++p) {
        const DevLink Lk = nxt;
        nxt = load_link(links, p + 1 < L ? p + 1 : p);
        walk_step<false>(hdr, Lk, p, qs, D, lane, slots, cur, par);
        const int col = sel.col[Lk.link];
        if (col < 0) continue;
        if (MODE == 0) {
            // A lane writing its own 64-byte matrix with four 16-byte stores makes every store instruction touch 64
            // different lines.  Through LDS instead: 4 lanes cover one sample's matrix, so an instruction writes 16
            // whole 64-byte blocks (4x fewer line visits; stride 20 floats keeps ds_*_b128 aligned and spread).
            float4* st = reinterpret_cast<float4*>(tile + lane * 20);
            st[0] = make_float4(cur.r[0], cur.r[1], cur.r[2], cur.t[0]);
            st[1] = make_float4(cur.r[3], cur.r[4], cur.r[5], cur.t[1]);
            st[2] = make_float4(cur.r[6], cur.r[7], cur.r[8], cur.t[2]);
            st[3] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            __syncthreads();
            const int sub = lane & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = (lane >> 2) + 16 * j;
                if (sl < rows)
                    *reinterpret_cast<float4*>(out + ((base + sl) * n_sel + col) * 16 + 4 * sub) =
                        *reinterpret_cast<const float4*>(tile + sl * 20 + 4 * sub);
            }
            __syncthreads();
        } else {
            float* o = tile + lane * rs + 3 * col;
            o[0] = cur.t[0]; o[1] = cur.t[1]; o[2] = cur.t[2];
        }
    }
    if (MODE == 1) {
        __syncthreads();
        store_tile_strided(out, tile, base * width, rows, width, rs, lane);
    }
}

// ============================================================================================
// Reverse pass shared by fk_backward and the fused rollout.
// ADJ supplies (Rbar, tbar) of the link at a position.  LDS: q | gq | joint records [7][D][64] | slots.
// ============================================================================================
struct JointRec { float a[3], m[3], c; };

template <bool PREFETCH = true, class ADJ>      // PREFETCH: request link record p + 1 while working on p (32 more live SGPRs)
__device__ __forceinline__ void reverse_walk(const DevModelHdr& hdr, const DevLink* __restrict__ links,
                                             const int32_t* __restrict__ fin, const float* qs, float* gqs,
                                             float* jst, float* slots, int lane, ADJ& adj) {
    const int D = hdr.n_dofs, L = hdr.n_links;
    float Pf[3] = {0.0f, 0.0f, 0.0f}, Pt[3] = {0.0f, 0.0f, 0.0f};
    Pose cur, par;
    DevLink nxt;
    if (PREFETCH) nxt = load_link(links, 0);
    adj.prefetch(0);
    for (int p = 0; p < L; ++p) {
        DevLink Lk;
        if (PREFETCH) { Lk = nxt; nxt = load_link(links, p + 1 < L ? p + 1 : p); }
        else Lk = load_link(links, p);
        const float pass = walk_step<false>(hdr, Lk, p, qs, D, lane, slots, cur, par);
        if (Lk.dof >= 0) {
            float a[3] = {0.0f, 0.0f, 0.0f}, m[3];
            if (Lk.type == TRK_JOINT_PRISMATIC) {
                // displacement direction R_parent * axis (axis is not rotated by R_fixed, rigid_body.py:176)
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    m[r] = -pass * fmaf(par.r[3 * r], Lk.axis[0], fmaf(par.r[3 * r + 1], Lk.axis[1], par.r[3 * r + 2] * Lk.axis[2]));
            } else {
                const float sg = pass * Lk.rot_sign;
                const int ax = Lk.rot_axis;
                const float z0 = ax == 0 ? cur.r[0] : (ax == 1 ? cur.r[1] : cur.r[2]);
                const float z1 = ax == 0 ? cur.r[3] : (ax == 1 ? cur.r[4] : cur.r[5]);
                const float z2 = ax == 0 ? cur.r[6] : (ax == 1 ? cur.r[7] : cur.r[8]);
                a[0] = sg * z0; a[1] = sg * z1; a[2] = sg * z2;
                m[0] = a[1] * cur.t[2] - a[2] * cur.t[1];
                m[1] = a[2] * cur.t[0] - a[0] * cur.t[2];
                m[2] = a[0] * cur.t[1] - a[1] * cur.t[0];
            }
            const float c = (a[0] * Pt[0] + a[1] * Pt[1] + a[2] * Pt[2]) - (m[0] * Pf[0] + m[1] * Pf[1] + m[2] * Pf[2]);
            float* j = jst + Lk.dof * TRK_WAVE + lane;
            const int js = D * TRK_WAVE;
            j[0] = a[0]; j[js] = a[1]; j[2 * js] = a[2];
            j[3 * js] = m[0]; j[4 * js] = m[1]; j[5 * js] = m[2];
            gqs[lane * D + Lk.dof] = c;         // the snapshot waits in the gradient's own slot until the joint's subtree ends
        }
        // own wrench
        float Rb[9], tb[3];
        if (adj(Lk, p, cur, Rb, tb)) {
            Pf[0] += tb[0]; Pf[1] += tb[1]; Pf[2] += tb[2];
            Pt[0] += cur.t[1] * tb[2] - cur.t[2] * tb[1];
            Pt[1] += cur.t[2] * tb[0] - cur.t[0] * tb[2];
            Pt[2] += cur.t[0] * tb[1] - cur.t[1] * tb[0];
            if (adj.has_rot(Lk)) {
                // M = Rbar R^T ; torque = (M21 - M12, M02 - M20, M10 - M01)
                const float M21 = Rb[6] * cur.r[3] + Rb[7] * cur.r[4] + Rb[8] * cur.r[5];
                const float M12 = Rb[3] * cur.r[6] + Rb[4] * cur.r[7] + Rb[5] * cur.r[8];
                const float M02 = Rb[0] * cur.r[6] + Rb[1] * cur.r[7] + Rb[2] * cur.r[8];
                const float M20 = Rb[6] * cur.r[0] + Rb[7] * cur.r[1] + Rb[8] * cur.r[2];
                const float M10 = Rb[3] * cur.r[0] + Rb[4] * cur.r[1] + Rb[5] * cur.r[2];
                const float M01 = Rb[0] * cur.r[3] + Rb[1] * cur.r[4] + Rb[2] * cur.r[5];
                Pt[0] += M21 - M12; Pt[1] += M02 - M20; Pt[2] += M10 - M01;
            }
        }
        if (p + 1 < L) adj.prefetch(p + 1);       // the next position's adjoint travels while this one is finished
        // joints whose subtree ends here
        for (int f = Lk.fin_begin; f < Lk.fin_end; ++f) {
            const int d = fin[f];
            const float* j = jst + d * TRK_WAVE + lane;
            const int js = D * TRK_WAVE;
            const float g = (j[0] * Pt[0] + j[js] * Pt[1] + j[2 * js] * Pt[2]) -
                            (j[3 * js] * Pf[0] + j[4 * js] * Pf[1] + j[5 * js] * Pf[2]) - gqs[lane * D + d];
            gqs[lane * D + d] = g;
        }
    }
}

struct AdjFromGH {        // gH [N, n_sel, 4, 4]
    // Plain per-lane loads at the point of use.  Two "smarter" variants were measured slower on MI355X (4096 x 64,
    // Panda, 42 us as written): a cooperative 4-lanes-per-matrix fetch through LDS (the mirror of k_fk_forward's
    // store path; 77 us -- a full round trip between two barriers per link) and a one-position-ahead register
    // prefetch (69 us).
    const float* gH; int64_t s; int n_sel; const SelMap& sel; bool valid;
    __device__ __forceinline__ bool has_rot(const DevLink&) const { return true; }
    __device__ __forceinline__ void prefetch(int) const {}
    __device__ __forceinline__ bool operator()(const DevLink& Lk, int, const Pose&, float* Rb, float* tb) const {
        const int col = sel.col[Lk.link];
        if (col < 0) return false;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
        if (valid) {
            const float4* g = reinterpret_cast<const float4*>(gH + (s * n_sel + col) * 16);
            r0 = g[0]; r1 = g[1]; r2 = g[2];
        }
        Rb[0] = r0.x; Rb[1] = r0.y; Rb[2] = r0.z; tb[0] = r0.w;
        Rb[3] = r1.x; Rb[4] = r1.y; Rb[5] = r1.z; tb[1] = r1.w;
        Rb[6] = r2.x; Rb[7] = r2.y; Rb[8] = r2.z; tb[2] = r2.w;
        return true;
    }
};

template <bool IDENT>
struct AdjFromTile {      // tbar from an LDS tile [64][rs] (link-major, 3 floats per link) + optional EE rotation adjoints
    const float* tile; int rs; int lane; const SelMap& sel; int ee_link; const float* eeRb; int ee2_link = -1; const float* ee2Rb = nullptr;
    __device__ __forceinline__ bool has_rot(const DevLink& Lk) const { return Lk.link == ee_link || Lk.link == ee2_link; }
    __device__ __forceinline__ void prefetch(int) const {}
    __device__ __forceinline__ bool operator()(const DevLink& Lk, int, const Pose&, float* Rb, float* tb) const {
        const int col = IDENT ? Lk.link : sel.col[Lk.link];
        if (col < 0) return false;
        const float* g = tile + lane * rs + 3 * col;
        tb[0] = g[0]; tb[1] = g[1]; tb[2] = g[2];
        if (Lk.link == ee_link) {
#pragma unroll
            for (int k = 0; k < 9; ++k) Rb[k] = eeRb[k];
        } else if (Lk.link == ee2_link) {
#pragma unroll
            for (int k = 0; k < 9; ++k) Rb[k] = ee2Rb[k];
        }
        return true;
    }
};

// MODE 0: adjoint gH [N,n_sel,4,4]; MODE 1: adjoint gpos [N,n_sel,3].  -> gq [N,D]
template <int MODE>
__global__ void __launch_bounds__(TRK_WAVE)
k_fk_backward(DevModelHdr hdr, const DevLink* __restrict__ links, const int32_t* __restrict__ fin, SelMap sel, SelMap selp,
              int n_sel, const float* __restrict__ q, const float* __restrict__ gin, int64_t n, float* __restrict__ gq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* qs = smem;
    float* gqs = qs + TRK_WAVE * D;
    float* jst = gqs + TRK_WAVE * D;
    float* slots = jst + 6 * D * TRK_WAVE;
    float* tile = slots + hdr.n_slots * 12 * TRK_WAVE;
    const int width = n_sel * 3, rs = width | 1;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    for (int k = lane; k < TRK_WAVE * D; k += TRK_WAVE) gqs[k] = 0.0f;
    if (MODE == 1) {
        for (int k = lane; k < TRK_WAVE * rs; k += TRK_WAVE) tile[k] = 0.0f;
        __syncthreads();
        load_tile_strided(tile, gin, base * width, rows, width, rs, lane);
    }
    __syncthreads();
    if (MODE == 0) {
        AdjFromGH adj{gin, base + lane, n_sel, sel, lane < rows};
        reverse_walk(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
    } else {
        AdjFromTile<false> adj{tile, rs, lane, sel, -1, nullptr};
        reverse_walk(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
    }
    __syncthreads();
    store_tile(gq, gqs, base * D, (int64_t)rows * D, lane);
}

// ============================================================================================
// Attached points: world positions of points fixed in link frames, R_link * off + t_link
// (Frame.transform_point frame.py:116-118 as used by RobotPanda.fk_map_collision_impl robot_panda.py:154-168),
// and the explicit reverse mode.  A point adjoint g is the wrench (g, p x g) on its link:
// tbar = sum g, Rbar = sum g off^T  (the shared reverse walk turns Rbar R^T into the torque sum (R off) x g).
// LDS: q | pose slots | point tile [64][3P|1].
// ============================================================================================
__device__ __forceinline__ void points_of_link(const DevPointSet& ps, int p, const Pose& cur, float* row) {
    const int b = cptr(ps.begin)[p], e = cptr(ps.begin)[p + 1];
    for (int k = b; k < e; ++k) {
        const TRK_CAS DevPoint* pt = cptr(ps.pts) + k;
        const float c0 = pt->off[0], c1 = pt->off[1], c2 = pt->off[2];
        float* o = row + 3 * pt->col;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            o[r] = fmaf(cur.r[3 * r], c0, fmaf(cur.r[3 * r + 1], c1, fmaf(cur.r[3 * r + 2], c2, cur.t[r])));
    }
}

__global__ void __launch_bounds__(TRK_WAVE)
k_fk_points(DevModelHdr hdr, const DevLink* __restrict__ links, DevPointSet ps, const float* __restrict__ q, int64_t n,
            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs, L = hdr.n_links;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* qs = smem;
    float* slots = qs + TRK_WAVE * D;
    float* tile = slots + hdr.n_slots * 12 * TRK_WAVE;
    const int width = ps.n_points * 3, rs = width | 1;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    __syncthreads();
    Pose cur, par;
    DevLink nxt = load_link(links, 0);
    for (int p = 0; p < L; ++p) {
        const DevLink Lk = nxt;
        nxt = load_link(links, p + 1 < L ? p + 1 : p);
        walk_step<false>(hdr, Lk, p, qs, D, lane, slots, cur, par);
        points_of_link(ps, p, cur, tile + lane * rs);
    }
    __syncthreads();
    store_tile_strided(out, tile, base * width, rows, width, rs, lane);
}

struct AdjFromPoints {    // point adjoints from an LDS tile [64][rs] (+ optional EE pose adjoint on one link)
    const float* gtile; int rs; int lane; DevPointSet ps; int ee_link; const float* eeRb; const float* eetb;
    int ee2_link = -1; const float* ee2Rb = nullptr; const float* ee2tb = nullptr;
    __device__ __forceinline__ bool has_rot(const DevLink&) const { return true; }
    __device__ __forceinline__ void prefetch(int) const {}
    __device__ __forceinline__ bool operator()(const DevLink& Lk, int p, const Pose&, float* Rb, float* tb) const {
        const int b = cptr(ps.begin)[p], e = cptr(ps.begin)[p + 1];
        const bool ee = Lk.link == ee_link, ee2 = Lk.link == ee2_link;
        if (b == e && !ee && !ee2) return false;
#pragma unroll
        for (int k = 0; k < 9; ++k) Rb[k] = ee ? eeRb[k] : (ee2 ? ee2Rb[k] : 0.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) tb[k] = ee ? eetb[k] : (ee2 ? ee2tb[k] : 0.0f);
        for (int k = b; k < e; ++k) {
            const TRK_CAS DevPoint* pt = cptr(ps.pts) + k;
            const float c0 = pt->off[0], c1 = pt->off[1], c2 = pt->off[2];
            const float* g = gtile + lane * rs + 3 * pt->col;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float gr = g[r];
                tb[r] += gr;
                Rb[3 * r] = fmaf(gr, c0, Rb[3 * r]); Rb[3 * r + 1] = fmaf(gr, c1, Rb[3 * r + 1]); Rb[3 * r + 2] = fmaf(gr, c2, Rb[3 * r + 2]);
            }
        }
        return true;
    }
};

__global__ void __launch_bounds__(TRK_WAVE)
k_fk_points_backward(DevModelHdr hdr, const DevLink* __restrict__ links, const int32_t* __restrict__ fin, DevPointSet ps,
                     const float* __restrict__ q, const float* __restrict__ gin, int64_t n, float* __restrict__ gq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* qs = smem;
    float* gqs = qs + TRK_WAVE * D;
    float* jst = gqs + TRK_WAVE * D;
    float* slots = jst + 6 * D * TRK_WAVE;
    float* tile = slots + hdr.n_slots * 12 * TRK_WAVE;
    const int width = ps.n_points * 3, rs = width | 1;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    for (int k = lane; k < TRK_WAVE * D; k += TRK_WAVE) gqs[k] = 0.0f;
    for (int k = lane; k < TRK_WAVE * rs; k += TRK_WAVE) tile[k] = 0.0f;
    __syncthreads();
    load_tile_strided(tile, gin, base * width, rows, width, rs, lane);
    __syncthreads();
    AdjFromPoints adj{tile, rs, lane, ps, -1, nullptr, nullptr};
    reverse_walk(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
    __syncthreads();
    store_tile(gq, gqs, base * D, (int64_t)rows * D, lane);
}

// ============================================================================================
// One iteration of DifferentiableTree.inverse_kinematics (robot_tree.py:345-377) as ONE kernel: FK of the target link,
// SE3_distance + joint-limit hinge (loss_fn_ik_per_q :386-417), its gradient (reverse walk), the termination test
// (ik_termination :419-442, on q BEFORE the update) and the Adam update (torch.optim.Adam defaults), all per lane.
// The reference runs two FK passes, an autograd backward and ~10 optimizer kernels per iteration.
// ============================================================================================
struct AdjIK {            // adjoint of the SE(3) distance on one link, evaluated when the walk reaches it
    int link; const float* Ht; float err;
    __device__ __forceinline__ bool has_rot(const DevLink&) const { return true; }
    __device__ __forceinline__ void prefetch(int) const {}
    __device__ __forceinline__ bool operator()(const DevLink& Lk, int, const Pose& cur, float* Rb, float* tb) {
        if (Lk.link != link) return false;
        err = ee_cost_eval(cur.r, cur.t, Ht, 1.0f, 1.0f, 0, Rb, tb);
        return true;
    }
};

__global__ void __launch_bounds__(TRK_WAVE)
k_ik_step(DevModelHdr hdr, const DevLink* __restrict__ links, const int32_t* __restrict__ fin, int link,
          const float* __restrict__ H_target, int per_sample, const float* __restrict__ lower,
          const float* __restrict__ upper, float w_jl, float se3_eps, float lr, IkSchedule sched, int n_steps, int64_t n,
          float* __restrict__ q, float* __restrict__ mom, float* __restrict__ vel, float* __restrict__ loss,
          uint8_t* __restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* qs = smem;
    float* gqs = qs + TRK_WAVE * D;
    float* jst = gqs + TRK_WAVE * D;
    float* slots = jst + 6 * D * TRK_WAVE;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    float Ht[16];
    {
        const int64_t s = min(base + lane, n - 1);
        const float* tp = H_target + (per_sample ? s * 16 : 0);
#pragma unroll
        for (int k = 0; k < 12; ++k) Ht[k] = tp[k];
    }
    const int64_t s = base + lane;
    // n_steps iterations on the tile in LDS: the configurations never leave the chip between them (the Adam state does, but
    // every lane re-reads only what it wrote itself).  loss / valid describe q as it was when the launch started.
    for (int it = 0; it < n_steps; ++it) {
        for (int k = lane; k < TRK_WAVE * D; k += TRK_WAVE) gqs[k] = 0.0f;
        __syncthreads();
        AdjIK adj{link, Ht, 0.0f};
        reverse_walk(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
        // hinge on the (shrunk) joint limits, termination test, Adam; each lane owns its row of the q / gq tiles
        bool ok = adj.err < se3_eps;
        float jl = 0.0f;
        const float bc1 = sched.bc1[it], rsqrt_bc2 = sched.rsqrt_bc2[it];
        for (int d = 0; d < D; ++d) {
            const float qv = qs[lane * D + d], lo = lower[d], hi = upper[d];
            float g = gqs[lane * D + d];
            if (qv < lo) { const float e = lo - qv; jl = fmaf(e, e, jl); g = fmaf(-2.0f * w_jl, e, g); }
            if (qv > hi) { const float e = hi - qv; jl = fmaf(e, e, jl); g = fmaf(-2.0f * w_jl, e, g); }
            ok = ok && (qv >= lo) && (qv <= hi);
            if (lane < rows && lr > 0.0f) {
                const int64_t idx = s * D + d;
                const float m1 = fmaf(0.9f, mom[idx], 0.1f * g);
                const float v1 = fmaf(0.999f, vel[idx], 0.001f * g * g);
                mom[idx] = m1; vel[idx] = v1;
                const float denom = fmaf(sqrtf(v1), rsqrt_bc2, 1e-8f);
                qs[lane * D + d] = qv - (lr / bc1) * (m1 / denom);
            }
        }
        if (it == 0 && lane < rows) {
            if (loss) loss[s] = fmaf(w_jl, jl, adj.err);
            if (valid) valid[s] = ok ? 1 : 0;
        }
        __syncthreads();
    }
    if (lr > 0.0f) store_tile(q, qs, base * D, (int64_t)rows * D, lane);
}

// ============================================================================================
// Collision fields on link positions held in an LDS tile [64][rs].
// distance_fields.py:107-124 ('sdf'): objects :298-316, workspace box :319-332, self pairs :194-208.
// Returns the summed cost of the selected fields; if gtile != nullptr accumulates scale * d cost / d pos.
// ============================================================================================
// objects + workspace box for NB collision links [l0, l0 + NB) at once (independent chains for the scheduler)
template <int NB>
__device__ __forceinline__ float fields_links(const DevCostHdr& C, int fields, float w_obj, float w_ws, int l0,
                                              const float* pt, float* gt) {
    int li[NB];
    float mg[NB], x[NB], y[NB], z[NB], ax[NB], ay[NB], az[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        li[k] = cptr(C.obj_link_idx)[l0 + k];
        mg[k] = cptr(C.obj_link_margin)[l0 + k];
        x[k] = pt[3 * li[k]]; y[k] = pt[3 * li[k] + 1]; z[k] = pt[3 * li[k] + 2];
        ax[k] = 0.0f; ay[k] = 0.0f; az[k] = 0.0f;
    }
    float cost = 0.0f;
    if ((fields & TRK_FIELD_OBJECTS) && C.n_objects > 0) {
        float s[NB], gx[NB], gy[NB], gz[NB];
        scene_min_sdf<NB>(C, x, y, z, s, gx, gy, gz);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float v = mg[k] - s[k];
            const bool off = (C.clamp_fields & TRK_FIELD_OBJECTS) && !(v > 0.0f);       // clamp_sdf: relu(margin - sdf)
            const float wk = off ? 0.0f : w_obj;
            cost = fmaf(wk, v, cost);
            ax[k] -= wk * gx[k]; ay[k] -= wk * gy[k]; az[k] -= wk * gz[k];
        }
    }
    if ((fields & TRK_FIELD_WS) && C.has_ws) {
#pragma unroll
        for (int k = 0; k < NB; ++k) cost = fmaf(w_ws, ws_cost_point(C, mg[k], x[k], y[k], z[k], w_ws, ax[k], ay[k], az[k]), cost);
    }
    if (gt) {
#pragma unroll
        for (int k = 0; k < NB; ++k) { gt[3 * li[k]] += ax[k]; gt[3 * li[k] + 1] += ay[k]; gt[3 * li[k] + 2] += az[k]; }
    }
    return cost;
}

// boolean version of fields_links (distance_fields.py:210-215, 283-291): NB collision links [l0, l0 + NB) at once
template <int NB>
__device__ __forceinline__ bool collision_links(const DevCostHdr& C, int fields, int l0, float margin_override, int use_default,
                                                const float* pt) {
    float mg[NB], x[NB], y[NB], z[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int li = cptr(C.obj_link_idx)[l0 + k];
        mg[k] = use_default ? cptr(C.obj_link_margin)[l0 + k] : margin_override;
        x[k] = pt[3 * li]; y[k] = pt[3 * li + 1]; z[k] = pt[3 * li + 2];
    }
    bool hit = false;
    if ((fields & TRK_FIELD_OBJECTS) && C.n_objects > 0) {
        float s[NB], gx[NB], gy[NB], gz[NB];
        scene_min_sdf<NB>(C, x, y, z, s, gx, gy, gz);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            bool h = s[k] < mg[k];
            if (__builtin_fabsf(s[k] - mg[k]) < 1e-5f) {
                h = false;
                for (int o = 0; o < C.n_objects; ++o) {
                    float ax, ay, az;
                    h |= object_sdf<true>(C, o, x[k], y[k], z[k], ax, ay, az) < mg[k];
                }
            }
            hit |= h;
        }
    }
    if ((fields & TRK_FIELD_WS) && C.has_ws) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            hit |= (x[k] - C.ws_min[0] < mg[k]) | (y[k] - C.ws_min[1] < mg[k]) | (z[k] - C.ws_min[2] < mg[k]) |
                   (C.ws_max[0] - x[k] < mg[k]) | (C.ws_max[1] - y[k] < mg[k]) | (C.ws_max[2] - z[k] < mg[k]);
    }
    return hit;
}

// interpolate_link_pos (interpolate_points_v1 distance_fields.py:66-69, used at :145-147): the virtual columns behind the real
// ones, column n_links_in + v = w0 * column src0 + w1 * column src1.  Each lane works on its own row of the tile.
__device__ __forceinline__ void virtual_columns(const DevCostHdr& C, float* pt) {
    for (int v = 0; v < C.n_virtual; ++v) {
        const int a = cptr(C.virtual_src)[2 * v], b = cptr(C.virtual_src)[2 * v + 1];
        const float wa = cptr(C.virtual_w)[2 * v], wb = cptr(C.virtual_w)[2 * v + 1];
        float* o = pt + 3 * (C.n_links_in + v);
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = wa * pt[3 * a + k] + wb * pt[3 * b + k];
    }
}
// reverse mode: the adjoint of an interpolated point goes to its two source columns with its weights
__device__ __forceinline__ void virtual_columns_adjoint(const DevCostHdr& C, float* gt) {
    for (int v = 0; v < C.n_virtual; ++v) {
        const int a = cptr(C.virtual_src)[2 * v], b = cptr(C.virtual_src)[2 * v + 1];
        const float wa = cptr(C.virtual_w)[2 * v], wb = cptr(C.virtual_w)[2 * v + 1];
        const float* g = gt + 3 * (C.n_links_in + v);
#pragma unroll
        for (int k = 0; k < 3; ++k) { gt[3 * a + k] = fmaf(wa, g[k], gt[3 * a + k]); gt[3 * b + k] = fmaf(wb, g[k], gt[3 * b + k]); }
    }
}

template <bool PRECISE>
__device__ __forceinline__ float fields_eval(const DevCostHdr& C, int fields, float w_self, float w_obj, float w_ws,
                                             const float* tile, float* gtile, int rs, int lane) {
    float cost = 0.0f;
    const float* pt = tile + lane * rs;
    float* gt = gtile ? gtile + lane * rs : nullptr;
    if ((fields & (TRK_FIELD_OBJECTS | TRK_FIELD_WS)) != 0) {
        if (PRECISE) {
            for (int l = 0; l < C.n_obj_links; ++l) {
                const int li = cptr(C.obj_link_idx)[l];
                const float mg = cptr(C.obj_link_margin)[l];
                const float x = pt[3 * li], y = pt[3 * li + 1], z = pt[3 * li + 2];
                float ax = 0.0f, ay = 0.0f, az = 0.0f;
                if ((fields & TRK_FIELD_OBJECTS) && C.n_objects > 0) {
                    float best = 0.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
                    for (int o = 0; o < C.n_objects; ++o) {
                        float gx, gy, gz;
                        const float v = mg - object_sdf<true>(C, o, x, y, z, gx, gy, gz);
                        const bool take = (o == 0) || (v > best);       // max over objects, first maximum wins
                        best = take ? v : best; bx = take ? gx : bx; by = take ? gy : by; bz = take ? gz : bz;
                    }
                    const float wk = ((C.clamp_fields & TRK_FIELD_OBJECTS) && !(best > 0.0f)) ? 0.0f : w_obj;   // clamp_sdf
                    cost = fmaf(wk, best, cost);
                    ax -= wk * bx; ay -= wk * by; az -= wk * bz;
                }
                if ((fields & TRK_FIELD_WS) && C.has_ws) cost = fmaf(w_ws, ws_cost_point(C, mg, x, y, z, w_ws, ax, ay, az), cost);
                if (gt) { gt[3 * li] += ax; gt[3 * li + 1] += ay; gt[3 * li + 2] += az; }
            }
        } else {
            int l = 0;
            for (; l + 4 <= C.n_obj_links; l += 4) cost += fields_links<4>(C, fields, w_obj, w_ws, l, pt, gt);
            if (l + 2 <= C.n_obj_links) { cost += fields_links<2>(C, fields, w_obj, w_ws, l, pt, gt); l += 2; }
            if (l < C.n_obj_links) cost += fields_links<1>(C, fields, w_obj, w_ws, l, pt, gt);
        }
    }
    if (fields & TRK_FIELD_SELF) {
        for (int pi = 0; pi < C.n_self_pairs; ++pi) {
            const int a = cptr(C.self_pairs)[2 * pi], b = cptr(C.self_pairs)[2 * pi + 1];
            if (C.self_single && a == b) {      // one self-collision link: "distance" 1e9 |p|_1 (distance_fields.py:195-198)
                const float x = pt[3 * a], y = pt[3 * a + 1], z = pt[3 * a + 2];
                const float vs = cptr(C.self_margin)[pi] - ((__builtin_fabsf(x) + __builtin_fabsf(y)) + __builtin_fabsf(z)) * 1e9f;
                const float wsf = ((C.clamp_fields & TRK_FIELD_SELF) && !(vs > 0.0f)) ? 0.0f : w_self;
                cost = fmaf(wsf, vs, cost);
                if (gt) {       // d|x| / dx = sign(x), 0 at 0 (torch.abs)
                    const float k9 = wsf * 1e9f;
                    gt[3 * a] -= x > 0.0f ? k9 : (x < 0.0f ? -k9 : 0.0f);
                    gt[3 * a + 1] -= y > 0.0f ? k9 : (y < 0.0f ? -k9 : 0.0f);
                    gt[3 * a + 2] -= z > 0.0f ? k9 : (z < 0.0f ? -k9 : 0.0f);
                }
                continue;
            }
            const float dx = pt[3 * a] - pt[3 * b], dy = pt[3 * a + 1] - pt[3 * b + 1], dz = pt[3 * a + 2] - pt[3 * b + 2];
            const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            const float nrm = PRECISE ? sqrtf(n2) : trk_sqrt(n2);
            const float vs = cptr(C.self_margin)[pi] - nrm;
            const float wsf = ((C.clamp_fields & TRK_FIELD_SELF) && !(vs > 0.0f)) ? 0.0f : w_self;          // clamp_sdf
            cost = fmaf(wsf, vs, cost);
            if (gt) {
                const float inv = nrm > 0.0f ? wsf * (PRECISE ? 1.0f / nrm : trk_rcp(nrm)) : 0.0f;
                const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
                gt[3 * a] -= ux; gt[3 * a + 1] -= uy; gt[3 * a + 2] -= uz;
                gt[3 * b] += ux; gt[3 * b + 1] += uy; gt[3 * b + 2] += uz;
            }
        }
    }
    return cost;
}

// cost [N] and (nullable) g_link_pos [N, Lin, 3] = gcost[n] * d cost[n] / d link_pos
__global__ void __launch_bounds__(TRK_WAVE)
k_cost_fields(DevCostHdr C, int fields, const float* __restrict__ link_pos, int64_t n,
              const float* __restrict__ gcost, float* __restrict__ cost, float* __restrict__ g_link_pos) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int width = C.n_links_in * 3, rs = ((C.n_links_in + C.n_virtual) * 3) | 1;     // rows carry the virtual columns too
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* tile = smem;
    float* gtile = tile + TRK_WAVE * rs;                      // allocated only when g_link_pos != nullptr
    if (g_link_pos) for (int k = lane; k < TRK_WAVE * rs; k += TRK_WAVE) gtile[k] = 0.0f;
    if (rows < TRK_WAVE) for (int k = rows * rs + lane; k < TRK_WAVE * rs; k += TRK_WAVE) tile[k] = 0.0f;   // lanes past the end compute on zeros
    __syncthreads();
    load_tile_strided(tile, link_pos, base * width, rows, width, rs, lane);
    __syncthreads();
    const float sc = (gcost && lane < rows) ? gcost[base + lane] : 1.0f;
    if (C.n_virtual) virtual_columns(C, tile + lane * rs);
    const float c = fields_eval<false>(C, fields, 1.0f, 1.0f, 1.0f, tile, g_link_pos ? gtile : nullptr, rs, lane);
    if (lane < rows) cost[base + lane] = c;
    if (g_link_pos) {
        if (C.n_virtual) virtual_columns_adjoint(C, gtile + lane * rs);
        if (gcost) for (int k = 0; k < width; ++k) gtile[lane * rs + k] *= sc;
        __syncthreads();
        store_tile_strided(g_link_pos, gtile, base * width, rows, width, rs, lane);
    }
}

// distance_fields.py:210-215, 283-291 (field_type='occupancy'); tasks.py:227-228 ORs the fields
__global__ void __launch_bounds__(TRK_WAVE)
k_collision_fields(DevCostHdr C, int fields, const float* __restrict__ link_pos, int64_t n,
                   float margin_override, int use_default, uint8_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int width = C.n_links_in * 3, rs = ((C.n_links_in + C.n_virtual) * 3) | 1;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    float* tile = smem;
    for (int k = lane; k < TRK_WAVE * rs; k += TRK_WAVE) tile[k] = 0.0f;
    __syncthreads();
    load_tile_strided(tile, link_pos, base * width, rows, width, rs, lane);
    __syncthreads();
    if (C.n_virtual) virtual_columns(C, tile + lane * rs);
    const float* pt = tile + lane * rs;
    bool hit = false;
    if (fields & (TRK_FIELD_OBJECTS | TRK_FIELD_WS)) {
        // four links at a time through the scene's minimum distance (the ranking of the cost kernels: one rsq per link), exact
        // object-by-object re-evaluation only for a lane whose distance lies within 1e-5 of its margin
        int l = 0;
        for (; l + 4 <= C.n_obj_links; l += 4) hit |= collision_links<4>(C, fields, l, margin_override, use_default, pt);
        if (l + 2 <= C.n_obj_links) { hit |= collision_links<2>(C, fields, l, margin_override, use_default, pt); l += 2; }
        if (l < C.n_obj_links) hit |= collision_links<1>(C, fields, l, margin_override, use_default, pt);
    }
    if (fields & TRK_FIELD_SELF) {
        for (int pi = 0; pi < C.n_self_pairs; ++pi) {
            const int a = cptr(C.self_pairs)[2 * pi], b = cptr(C.self_pairs)[2 * pi + 1];
            const float dx = pt[3 * a] - pt[3 * b], dy = pt[3 * a + 1] - pt[3 * b + 1], dz = pt[3 * a + 2] - pt[3 * b + 2];
            const float mg = use_default ? cptr(C.self_margin)[pi] : margin_override;
            if (C.self_single && a == b) {      // distance_fields.py:195-198
                hit |= ((__builtin_fabsf(pt[3 * a]) + __builtin_fabsf(pt[3 * a + 1])) + __builtin_fabsf(pt[3 * a + 2])) * 1e9f < mg;
                continue;
            }
            hit |= sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz))) < mg;
        }
    }
    if (lane < rows) out[base + lane] = hit ? 1 : 0;
}

// EESE3DistanceField.compute_costs_impl distance_fields.py:347-356
__global__ void __launch_bounds__(256)
k_ee_cost(DevCostHdr C, const float* __restrict__ H, int64_t n, int64_t stride, const float* __restrict__ target,
          int per_sample, const float* __restrict__ gcost, float* __restrict__ cost, float* __restrict__ gH, int64_t g_stride) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float4* h = reinterpret_cast<const float4*>(H + s * stride);
    const float4 h0 = h[0], h1 = h[1], h2 = h[2];
    const float R[9] = {h0.x, h0.y, h0.z, h1.x, h1.y, h1.z, h2.x, h2.y, h2.z};
    const float t[3] = {h0.w, h1.w, h2.w};
    float Ht[16];
    if (target) {
        const float* tp = target + (per_sample ? s * 16 : 0);
#pragma unroll
        for (int k = 0; k < 12; ++k) Ht[k] = tp[k];
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) Ht[k] = C.ee_target[k];
    }
    float gR[9], gt[3];
    const float c = ee_cost_eval(R, t, Ht, C.ee_w_pos, C.ee_w_rot, C.ee_square, gR, gt);
    cost[s] = c;
    if (gH) {
        const float sc = gcost ? gcost[s] : 1.0f;
        float4* g = reinterpret_cast<float4*>(gH + s * g_stride);
        g[0] = make_float4(sc * gR[0], sc * gR[1], sc * gR[2], sc * gt[0]);
        g[1] = make_float4(sc * gR[3], sc * gR[4], sc * gR[5], sc * gt[1]);
        g[2] = make_float4(sc * gR[6], sc * gR[7], sc * gR[8], sc * gt[2]);
        g[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);           // the bottom row of H is constant: the caller need not pre-zero gH
    }
}

// ============================================================================================
// Fused rollout, table-driven: walk 1 (FK -> position tile, EE rotation), costs + position adjoints,
// walk 2 (reverse pass).  q [N,D] -> link_pos [N,L,3] (nullable), cost [N], gq [N,D], cost_sum (nullable).
// ============================================================================================
// IO: HBM-side type of q / link_pos; G: of the gradient (fp16 q: multiplied by grad_scale, fp16 stores saturate)
template <bool POINTS, class IO, class G = IO>     // POINTS: the cost model's columns are the attached points of `ps`, not the links
__global__ void __launch_bounds__(TRK_WAVE)
k_rollout_generic(DevModelHdr hdr, const DevLink* __restrict__ links, const int32_t* __restrict__ fin, SelMap sel_unused, DevPointSet ps, DevCostHdr C,
                  TrkRolloutWeights w, const IO* __restrict__ q, int64_t n, IO* __restrict__ link_pos,
                  float* __restrict__ cost, G* __restrict__ gq, float* __restrict__ cost_sum, float grad_scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs, L = hdr.n_links;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    const int width = (POINTS ? ps.n_points : L) * 3, rs = (width + C.n_virtual * 3) | 1;   // + the interpolated columns
    float* qs = smem;
    float* gqs = qs + TRK_WAVE * D;
    float* jst = gqs + TRK_WAVE * D;
    float* slots = jst + 6 * D * TRK_WAVE;
    float* tile = slots + hdr.n_slots * 12 * TRK_WAVE;
    float* gtile = tile + TRK_WAVE * rs;
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    for (int k = lane; k < TRK_WAVE * D; k += TRK_WAVE) gqs[k] = 0.0f;
    for (int k = lane; k < TRK_WAVE * rs; k += TRK_WAVE) gtile[k] = 0.0f;
    __syncthreads();
    // walk 1
    float eeR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, eet[3] = {0, 0, 0};
    float e2R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, e2t[3] = {0, 0, 0};
    {
        Pose cur, par;
        for (int p = 0; p < L; ++p) {
            // no record prefetch in this kernel: two records in flight are 64 SGPRs, and with the cost model's header next to
            // them the kernel spilled ~170 scalars through v_writelane / v_readlane (103 -> 97 us without)
            const DevLink Lk = load_link(links, p);
            walk_step<false>(hdr, Lk, p, qs, D, lane, slots, cur, par);
            if (POINTS) points_of_link(ps, p, cur, tile + lane * rs);
            else {
                float* o = tile + lane * rs + 3 * Lk.link;
                o[0] = cur.t[0]; o[1] = cur.t[1]; o[2] = cur.t[2];
            }
            if (Lk.link == C.ee_link) {
#pragma unroll
                for (int k = 0; k < 9; ++k) eeR[k] = cur.r[k];
                eet[0] = cur.t[0]; eet[1] = cur.t[1]; eet[2] = cur.t[2];
            }
            if (Lk.link == C.ee2_link) {
#pragma unroll
                for (int k = 0; k < 9; ++k) e2R[k] = cur.r[k];
                e2t[0] = cur.t[0]; e2t[1] = cur.t[1]; e2t[2] = cur.t[2];
            }
        }
    }
    // costs and adjoints w.r.t. link positions (per-lane rows of the tiles: no cross-lane hazard)
    int fields = 0;
    if (w.w_self != 0.0f) fields |= TRK_FIELD_SELF;
    if (w.w_obj != 0.0f) fields |= TRK_FIELD_OBJECTS;
    if (w.w_ws != 0.0f) fields |= TRK_FIELD_WS;
    if (C.n_virtual) virtual_columns(C, tile + lane * rs);
    float c = fields_eval<false>(C, fields, w.w_self, w.w_obj, w.w_ws, tile, gtile, rs, lane);
    if (C.n_virtual) virtual_columns_adjoint(C, gtile + lane * rs);
    float eeRb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, eetb[3] = {0, 0, 0};
    const bool use_ee = (w.w_ee != 0.0f) && (C.ee_link >= 0);
    if (use_ee) {
        float gt[3];
        const float ce = ee_cost_eval(eeR, eet, C.ee_target, C.ee_w_pos, C.ee_w_rot, C.ee_square, eeRb, gt);
        c = fmaf(w.w_ee, ce, c);
#pragma unroll
        for (int k = 0; k < 9; ++k) eeRb[k] *= w.w_ee;
        if (POINTS) { eetb[0] = w.w_ee * gt[0]; eetb[1] = w.w_ee * gt[1]; eetb[2] = w.w_ee * gt[2]; }
        else {
            float* g = gtile + lane * rs + 3 * C.ee_link;
            g[0] = fmaf(w.w_ee, gt[0], g[0]); g[1] = fmaf(w.w_ee, gt[1], g[1]); g[2] = fmaf(w.w_ee, gt[2], g[2]);
        }
    }
    float e2Rb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, e2tb[3] = {0, 0, 0};
    const bool use_ee2 = use_ee && (C.ee2_link >= 0);
    if (use_ee2) {      // second tracked link (two-arm scenes): same weights, its own target
        float gt[3];
        const float ce = ee_cost_eval(e2R, e2t, C.ee2_target, C.ee_w_pos, C.ee_w_rot, C.ee_square, e2Rb, gt);
        c = fmaf(w.w_ee, ce, c);
#pragma unroll
        for (int k = 0; k < 9; ++k) e2Rb[k] *= w.w_ee;
        if (POINTS) { e2tb[0] = w.w_ee * gt[0]; e2tb[1] = w.w_ee * gt[1]; e2tb[2] = w.w_ee * gt[2]; }
        else {
            float* g = gtile + lane * rs + 3 * C.ee2_link;
            g[0] = fmaf(w.w_ee, gt[0], g[0]); g[1] = fmaf(w.w_ee, gt[1], g[1]); g[2] = fmaf(w.w_ee, gt[2], g[2]);
        }
    }
    if (lane < rows) cost[base + lane] = c;
    if (cost_sum) {     // per-wavefront partial sum: no atomics (4096 same-address atomics cost ~47 us on MI355X)
        const float tot = wave_sum(lane < rows ? c : 0.0f);
        if (lane == 0) cost_sum[blockIdx.x] = tot;
    }
    // walk 2: reverse pass
    if (POINTS) {
        AdjFromPoints adj{gtile, rs, lane, ps, use_ee ? C.ee_link : -1, eeRb, eetb, use_ee2 ? C.ee2_link : -1, e2Rb, e2tb};
        reverse_walk<false>(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
    } else {
        AdjFromTile<true> adj{gtile, rs, lane, sel_unused, use_ee ? C.ee_link : -1, eeRb, use_ee2 ? C.ee2_link : -1, e2Rb};
        reverse_walk<false>(hdr, links, fin, qs, gqs, jst, slots, lane, adj);
    }
    __syncthreads();
    if constexpr (std::is_same<IO, float>::value) store_tile(gq, gqs, base * D, (int64_t)rows * D, lane);
    else store_tile_scaled(gq, gqs, base * D, (int64_t)rows * D, lane, grad_scale);
    if (link_pos) store_tile_strided(link_pos, tile, base * width, rows, width, rs, lane);
}

// ============================================================================================
// Stateful FK + geometric Jacobian (robot_tree.py:136-190, 218-248; Frame.get_quaternion frame.py:87-114)
// LDS: q | qd | slots (pose 12 + velocity 6 floats per slot) | joint axes/origins [6][D][64].
// ============================================================================================
struct JacCols {                // DOF -> record slot of the Jacobian columns this call produces (-1: column stays zero)
    int8_t slot[TRK_MAX_DOFS];
    int32_t n_cols;
    int32_t p_end;              // the walk may stop after this pre-order position
};

// The column records (z, p) of the contributing joints live in LDS lane-major with an odd stride, so both the walk's
// per-lane writes and the final transposed reads are bank-conflict free; lin_jac / ang_jac [N,3,D] leave as contiguous
// 64*3D-float runs per wavefront (a per-lane store of 3D floats at a 12D-byte stride touches 64 lines per instruction:
// measured 361 us at 4096 x 64 on UR10+Allegro before this layout).
__global__ void __launch_bounds__(TRK_WAVE)
k_fk_jacobian(DevModelHdr hdr, const DevLink* __restrict__ links, JacCols cols, const float* __restrict__ q,
              const float* __restrict__ qd, int64_t n, int link, int link_joint_idx, float* __restrict__ pos,
              float* __restrict__ quat, float* __restrict__ lin_jac, float* __restrict__ ang_jac,
              float* __restrict__ vel_lin, float* __restrict__ vel_ang) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    const int rstride = (6 * cols.n_cols + 3) | 1;          // per lane: n_cols x (z, p) + the link position
    float* qs = smem;
    float* qds = qs + TRK_WAVE * D;                         // only when qd != nullptr
    float* slots = qds + (qd ? TRK_WAVE * D : 0);
    float* vslots = slots + hdr.n_slots * 12 * TRK_WAVE;
    float* rec = vslots + (qd ? hdr.n_slots * 6 * TRK_WAVE : 0);
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    if (qd) {
        load_tile(qds, qd, base * D, (int64_t)rows * D, lane);
        for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qds[k] = 0.0f;
    }
    float* myrec = rec + lane * rstride;
    for (int k = 0; k < rstride; ++k) myrec[k] = 0.0f;
    // DOF -> column slot, in LDS: the read-out below indexes it per lane, and a per-lane index into a by-value kernel
    // argument array goes through scratch memory
    int* slot_lds = reinterpret_cast<int*>(rec + TRK_WAVE * rstride);
#pragma unroll
    for (int d = 0; d < TRK_MAX_DOFS; ++d)
        if (lane == d) slot_lds[d] = cols.slot[d];
    __syncthreads();
    Pose cur, par;
    float vl[3] = {0, 0, 0}, va[3] = {0, 0, 0};
    float eR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, et[3] = {0, 0, 0}, evl[3] = {0, 0, 0}, eva[3] = {0, 0, 0};
    DevLink nxt = load_link(links, 0);
    for (int p = 0; p < cols.p_end; ++p) {
        const DevLink Lk = nxt;
        nxt = load_link(links, p + 1 < cols.p_end ? p + 1 : p);
        walk_step<true>(hdr, Lk, p, qs, D, lane, slots, cur, par);
        if (qd) {
            float pvl[3], pva[3];
            if (p > 0 && Lk.parent_slot >= 0) {
                const float* v = vslots + Lk.parent_slot * 6 * TRK_WAVE + lane;
#pragma unroll
                for (int k = 0; k < 3; ++k) { pvl[k] = v[k * TRK_WAVE]; pva[k] = v[(3 + k) * TRK_WAVE]; }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { pvl[k] = vl[k]; pva[k] = va[k]; }
            }
            if (p == 0) { vl[0] = vl[1] = vl[2] = 0.0f; va[0] = va[1] = va[2] = 0.0f; }
            else {
                // joint pose J = par^-1 o cur ; parentToChild = J^-1 : R = J.R^T, t = -J.R^T J.t  (frame.py:57-62)
                // J.R^T = cur.R^T par.R ; J.t = par.R^T (cur.t - par.t)  =>  t_inv = -cur.R^T (cur.t - par.t)
                float Rt[9];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Rt[3 * r + c] = cur.r[r] * par.r[c] + cur.r[3 + r] * par.r[3 + c] + cur.r[6 + r] * par.r[6 + c];
                const float d0 = cur.t[0] - par.t[0], d1 = cur.t[1] - par.t[1], d2 = cur.t[2] - par.t[2];
                float ti[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) ti[r] = -(cur.r[r] * d0 + cur.r[3 + r] * d1 + cur.r[6 + r] * d2);
                float Ra[3], Rl[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    Ra[r] = Rt[3 * r] * pva[0] + Rt[3 * r + 1] * pva[1] + Rt[3 * r + 2] * pva[2];
                    Rl[r] = Rt[3 * r] * pvl[0] + Rt[3 * r + 1] * pvl[1] + Rt[3 * r + 2] * pvl[2];
                }
                const float qdv = Lk.dof >= 0 ? qds[lane * D + Lk.dof] : 0.0f;
                vl[0] = ti[1] * Ra[2] - ti[2] * Ra[1] + Rl[0];
                vl[1] = ti[2] * Ra[0] - ti[0] * Ra[2] + Rl[1];
                vl[2] = ti[0] * Ra[1] - ti[1] * Ra[0] + Rl[2];
                va[0] = Ra[0] + qdv * Lk.axis[0]; va[1] = Ra[1] + qdv * Lk.axis[1]; va[2] = Ra[2] + qdv * Lk.axis[2];
            }
            if (Lk.store_slot >= 0) {
                float* v = vslots + Lk.store_slot * 6 * TRK_WAVE + lane;
#pragma unroll
                for (int k = 0; k < 3; ++k) { v[k * TRK_WAVE] = vl[k]; v[(3 + k) * TRK_WAVE] = va[k]; }
            }
        }
        if (Lk.dof >= 0 && (Lk.link - 1) <= link_joint_idx && Lk.jac_axis >= 0) {   // robot_tree.py:239-244
            const int ax = Lk.jac_axis;
            float* j = myrec + 6 * slot_lds[Lk.dof];
            j[0] = ax == 0 ? cur.r[0] : (ax == 1 ? cur.r[1] : cur.r[2]);
            j[1] = ax == 0 ? cur.r[3] : (ax == 1 ? cur.r[4] : cur.r[5]);
            j[2] = ax == 0 ? cur.r[6] : (ax == 1 ? cur.r[7] : cur.r[8]);
            j[3] = cur.t[0]; j[4] = cur.t[1]; j[5] = cur.t[2];
        }
        if (Lk.link == link) {
#pragma unroll
            for (int k = 0; k < 9; ++k) eR[k] = cur.r[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { et[k] = cur.t[k]; evl[k] = vl[k]; eva[k] = va[k]; }
        }
    }
    myrec[6 * cols.n_cols] = et[0]; myrec[6 * cols.n_cols + 1] = et[1]; myrec[6 * cols.n_cols + 2] = et[2];
    __syncthreads();
    trk_jac_readout(rec, slot_lds, rstride, cols.n_cols, D, rows, lin_jac + base * 3 * D, ang_jac + base * 3 * D, lane);
    if (lane >= rows) return;
    const int64_t s = base + lane;
    pos[s * 3] = et[0]; pos[s * 3 + 1] = et[1]; pos[s * 3 + 2] = et[2];
    float qo[4];
    frame_quat_wxyz(eR, qo);
    *reinterpret_cast<float4*>(quat + s * 4) = make_float4(qo[0], qo[1], qo[2], qo[3]);
    if (vel_lin) { vel_lin[s * 3] = evl[0]; vel_lin[s * 3 + 1] = evl[1]; vel_lin[s * 3 + 2] = evl[2]; }
    if (vel_ang) { vel_ang[s * 3] = eva[0]; vel_ang[s * 3 + 1] = eva[1]; vel_ang[s * 3 + 2] = eva[2]; }
}

// ============================================================================================
// compute_analytical_jacobian_all_links (robot_tree.py:250-265): J [N, L, 7, D] = d [pos, quat_wxyz] / d q of every
// link.  The reference runs autograd 7L times; here every joint leaves a record (omega = pass*sign*z_j, p_j) in LDS
// as the walk passes it, and each link combines the records of its ancestors (pre-order range test):
//   d p_i = omega x (p_i - p_j)   |   prismatic: pass * R_parent axis
//   d R_i = [omega]x R_i  ->  d quat through the selected candidate of rotation_matrix_to_q (quaternion.py:135-166)
// LDS: q | slots | joint records [6][D][64] | one link's 7xD block per lane (written out as contiguous 7D-float runs).
// ============================================================================================
struct DofRec { int32_t pos, end, type, _pad; };   // pre-order position of the joint's link, end of its subtree

__global__ void __launch_bounds__(TRK_WAVE)
k_fk_analytic_jacobian(DevModelHdr hdr, const DevLink* __restrict__ links, const DofRec* __restrict__ dofs,
                       const float* __restrict__ q, int64_t n, float* __restrict__ J) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int D = hdr.n_dofs, L = hdr.n_links;
    const int64_t base = (int64_t)blockIdx.x * TRK_WAVE;
    const int rows = (int)min((int64_t)TRK_WAVE, n - base);
    const int W = 7 * D, rs = W | 1, js = D * TRK_WAVE;
    float* qs = smem;
    float* slots = qs + TRK_WAVE * D;
    float* jrec = slots + hdr.n_slots * 12 * TRK_WAVE;      // [6][D][64]
    float* tile = jrec + 6 * D * TRK_WAVE;                  // [64][rs]
    load_tile(qs, q, base * D, (int64_t)rows * D, lane);
    for (int k = rows * D + lane; k < TRK_WAVE * D; k += TRK_WAVE) qs[k] = 0.0f;
    __syncthreads();
    Pose cur, par;
    DevLink nxt = load_link(links, 0);
    for (int p = 0; p < L; ++p) {
        const DevLink Lk = nxt;
        nxt = load_link(links, p + 1 < L ? p + 1 : p);
        const float pass = walk_step<false>(hdr, Lk, p, qs, D, lane, slots, cur, par);
        if (Lk.dof >= 0) {
            float* j = jrec + Lk.dof * TRK_WAVE + lane;
            if (Lk.type == TRK_JOINT_PRISMATIC) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    j[r * js] = pass * fmaf(par.r[3 * r], Lk.axis[0], fmaf(par.r[3 * r + 1], Lk.axis[1], par.r[3 * r + 2] * Lk.axis[2]));
            } else {
                const float sg = pass * Lk.rot_sign;
                const int ax = Lk.rot_axis;
                j[0] = sg * (ax == 0 ? cur.r[0] : (ax == 1 ? cur.r[1] : cur.r[2]));
                j[js] = sg * (ax == 0 ? cur.r[3] : (ax == 1 ? cur.r[4] : cur.r[5]));
                j[2 * js] = sg * (ax == 0 ? cur.r[6] : (ax == 1 ? cur.r[7] : cur.r[8]));
            }
            j[3 * js] = cur.t[0]; j[4 * js] = cur.t[1]; j[5 * js] = cur.t[2];
        }
        float* row = tile + lane * rs;
        for (int d = 0; d < D; ++d) {
            const DofRec T = dofs[d];
            float dp[3] = {0.0f, 0.0f, 0.0f}, dq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (T.pos <= p && p < T.end) {                  // joint d is on the path root..this link
                const float* j = jrec + d * TRK_WAVE + lane;
                const float w0 = j[0], w1 = j[js], w2 = j[2 * js];
                if (T.type == TRK_JOINT_PRISMATIC) {
                    dp[0] = w0; dp[1] = w1; dp[2] = w2;    // rotation unchanged: d quat = 0
                } else {
                    const float r0 = cur.t[0] - j[3 * js], r1 = cur.t[1] - j[4 * js], r2 = cur.t[2] - j[5 * js];
                    dp[0] = w1 * r2 - w2 * r1; dp[1] = w2 * r0 - w0 * r2; dp[2] = w0 * r1 - w1 * r0;
                    float dR[9];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v0 = cur.r[c], v1 = cur.r[3 + c], v2 = cur.r[6 + c];
                        dR[c] = w1 * v2 - w2 * v1; dR[3 + c] = w2 * v0 - w0 * v2; dR[6 + c] = w0 * v1 - w1 * v0;
                    }
                    quat_jvp(cur.r, dR, dq);
                }
            }
            row[d] = dp[0]; row[D + d] = dp[1]; row[2 * D + d] = dp[2];
            row[3 * D + d] = dq[0]; row[4 * D + d] = dq[1]; row[5 * D + d] = dq[2]; row[6 * D + d] = dq[3];
        }
        __syncthreads();
        // sample r of the wave owns the contiguous run J[base + r][link][:, :] of 7D floats.  Element e = lane + 64 i of the [rows][W] tile
        // -> (sample e / W, k = e % W): every lane stores, four LDS reads in flight per trip (round 6; before: one 49-lane store per
        // sample, 64 dependent read -> store trips per link: Panda 378 us for 572 MB at 4096 x 64)
        {
            const int total = rows * W;
            const float inv_W = 1.0f / (float)W;
            float* dst0 = J + (base * L + Lk.link) * W;
            const int64_t sstride = (int64_t)L * W;
            for (int e0 = lane; e0 < total; e0 += 4 * TRK_WAVE) {
                float v[4]; int sm[4], kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + TRK_WAVE * j;
                    sm[j] = trk_div_small(e < total ? e : 0, W, inv_W); kk[j] = (e < total ? e : 0) - sm[j] * W;
                    v[j] = tile[sm[j] * rs + kk[j]];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (e0 + TRK_WAVE * j < total) dst0[sm[j] * sstride + kk[j]] = v[j];
            }
        }
        __syncthreads();
    }
}

// rotation_matrix_to_q quaternion.py:135-166
__global__ void __launch_bounds__(256)
k_rotmat_to_quat(const float* __restrict__ R, int64_t n, int stride, int pitch, float* __restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float* m = R + s * stride;
    const float m00 = m[0], m01 = m[1], m02 = m[2];
    const float m10 = m[pitch], m11 = m[pitch + 1], m12 = m[pitch + 2];
    const float m20 = m[2 * pitch], m21 = m[2 * pitch + 1], m22 = m[2 * pitch + 2];
    const float a0 = 1.0f + m00 + m11 + m22, a1 = 1.0f + m00 - m11 - m22;
    const float a2 = 1.0f - m00 + m11 - m22, a3 = 1.0f - m00 - m11 + m22;
    const float q0 = a0 > 0.0f ? sqrtf(a0) : 0.0f, q1 = a1 > 0.0f ? sqrtf(a1) : 0.0f;
    const float q2 = a2 > 0.0f ? sqrtf(a2) : 0.0f, q3 = a3 > 0.0f ? sqrtf(a3) : 0.0f;
    int best = 0; float qb = q0;
    if (q1 > qb) { qb = q1; best = 1; }
    if (q2 > qb) { qb = q2; best = 2; }
    if (q3 > qb) { qb = q3; best = 3; }
    const float den = 2.0f * fmaxf(qb, 0.1f);
    float c0, c1, c2, c3;
    if (best == 0)      { c0 = q0 * q0; c1 = m21 - m12; c2 = m02 - m20; c3 = m10 - m01; }
    else if (best == 1) { c0 = m21 - m12; c1 = q1 * q1; c2 = m10 + m01; c3 = m02 + m20; }
    else if (best == 2) { c0 = m02 - m20; c1 = m10 + m01; c2 = q2 * q2; c3 = m12 + m21; }
    else                { c0 = m10 - m01; c1 = m20 + m02; c2 = m21 + m12; c3 = q3 * q3; }
    float4* o = reinterpret_cast<float4*>(out + s * 4);
    *o = make_float4(c0 / den, c1 / den, c2 / den, c3 / den);
}

// ============================================================================================
// Frame algebra on packed poses (geometrics/frame.py:55-121): R [n,9] row-major, t [n,3]; a frame given once (n == 1)
// broadcasts.  One lane per pose; 48 bytes in / out per pose, so these are plain streaming kernels.
// ============================================================================================
struct Pose3 { float r[9], t[3]; };
__device__ __forceinline__ Pose3 pose_load(const float* R, const float* t, int64_t i) {
    Pose3 p;
#pragma unroll
    for (int k = 0; k < 9; ++k) p.r[k] = R[i * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.t[k] = t[i * 3 + k];
    return p;
}
// frame.py:57-62: (R^T, -(R^T t))
__device__ __forceinline__ Pose3 pose_inverse(const Pose3& a) {
    Pose3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o.r[3 * i + j] = a.r[3 * j + i];
        o.t[i] = -(a.r[i] * a.t[0] + a.r[3 + i] * a.t[1] + a.r[6 + i] * a.t[2]);
    }
    return o;
}
// geometrics/utils.py:11-17: (Ra Rb, Ra tb + ta)
__device__ __forceinline__ Pose3 pose_mul(const Pose3& a, const Pose3& b) {
    Pose3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.r[3 * i + j] = a.r[3 * i] * b.r[j] + a.r[3 * i + 1] * b.r[3 + j] + a.r[3 * i + 2] * b.r[6 + j];
        o.t[i] = (a.r[3 * i] * b.t[0] + a.r[3 * i + 1] * b.t[1] + a.r[3 * i + 2] * b.t[2]) + a.t[i];
    }
    return o;
}
// adjoint of pose_mul: (gRa, gta, gRb, gtb) from (gR, gt)
__device__ __forceinline__ void pose_mul_bwd(const Pose3& a, const Pose3& b, const Pose3& g, Pose3& ga, Pose3& gb) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // gRa[i][k] = sum_j gR[i][j] Rb[k][j] + gt[i] tb[k];   gRb[i][k] = sum_j Ra[j][i] gR[j][k]
            ga.r[3 * i + k] = g.r[3 * i] * b.r[3 * k] + g.r[3 * i + 1] * b.r[3 * k + 1] + g.r[3 * i + 2] * b.r[3 * k + 2] +
                              g.t[i] * b.t[k];
            gb.r[3 * i + k] = a.r[i] * g.r[k] + a.r[3 + i] * g.r[3 + k] + a.r[6 + i] * g.r[6 + k];
        }
        ga.t[i] = g.t[i];
        gb.t[i] = a.r[i] * g.t[0] + a.r[3 + i] * g.t[1] + a.r[6 + i] * g.t[2];
    }
}
// adjoint of pose_inverse
__device__ __forceinline__ void pose_inverse_bwd(const Pose3& a, const Pose3& g, Pose3& ga) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ga.r[3 * j + i] = g.r[3 * i + j] - a.t[j] * g.t[i];
        ga.t[j] = -(a.r[3 * j] * g.t[0] + a.r[3 * j + 1] * g.t[1] + a.r[3 * j + 2] * g.t[2]);
    }
}
__device__ __forceinline__ void pose_store(float* R, float* t, int64_t i, const Pose3& p) {
    if (R) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[i * 9 + k] = p.r[k];
    }
    if (t) {
#pragma unroll
        for (int k = 0; k < 3; ++k) t[i * 3 + k] = p.t[k];
    }
}

__global__ void __launch_bounds__(256)
k_frame_compose(int op, const float* __restrict__ Ra, const float* __restrict__ ta, int a_bcast, const float* __restrict__ Rb,
                const float* __restrict__ tb, int b_bcast, int64_t n, float* __restrict__ Ro, float* __restrict__ to) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const Pose3 a = pose_load(Ra, ta, a_bcast ? 0 : s);
    Pose3 o;
    if (op == TRK_FRAME_INVERSE) o = pose_inverse(a);
    else {
        const Pose3 b = pose_load(Rb, tb, b_bcast ? 0 : s);
        o = op == TRK_FRAME_COMPOSE ? pose_mul(a, b) : pose_mul(pose_inverse(b), a);
    }
    pose_store(Ro, to, s, o);
}

__global__ void __launch_bounds__(256)
k_frame_compose_bwd(int op, const float* __restrict__ Ra, const float* __restrict__ ta, const float* __restrict__ Rb,
                    const float* __restrict__ tb, const float* __restrict__ gR, const float* __restrict__ gt, int64_t n,
                    float* __restrict__ gRa, float* __restrict__ gta, float* __restrict__ gRb, float* __restrict__ gtb) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const Pose3 a = pose_load(Ra, ta, s), g = pose_load(gR, gt, s);
    Pose3 ga, gb;
    if (op == TRK_FRAME_INVERSE) {
        pose_inverse_bwd(a, g, ga);
        pose_store(gRa, gta, s, ga);
        return;
    }
    const Pose3 b = pose_load(Rb, tb, s);
    if (op == TRK_FRAME_COMPOSE) pose_mul_bwd(a, b, g, ga, gb);
    else {                                     // out = inv(b) o a
        Pose3 gib;
        pose_mul_bwd(pose_inverse(b), a, g, gib, ga);
        pose_inverse_bwd(b, gib, gb);
    }
    pose_store(gRa, gta, s, ga);
    pose_store(gRb, gtb, s, gb);
}

// frame.py:116-118: out[s, p, :] = R_s point_p + t_s.  One workgroup row per pose block; lanes sweep the (pose, point)
// pairs so the [n, P, 3] output leaves as contiguous runs.
__global__ void __launch_bounds__(256)
k_frame_transform_points(const float* __restrict__ R, const float* __restrict__ t, int64_t n, const float* __restrict__ pts,
                         int P, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * P) return;
    const int64_t s = idx / P;
    const int p = (int)(idx - s * P);
    const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
    const float* r = R + s * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[idx * 3 + i] = (r[3 * i] * x + r[3 * i + 1] * y + r[3 * i + 2] * z) + t[s * 3 + i];
}
// reverse mode w.r.t. the pose: gR = sum_p g_p point_p^T, gt = sum_p g_p
__global__ void __launch_bounds__(256)
k_frame_transform_points_bwd(const float* __restrict__ g, int64_t n, const float* __restrict__ pts, int P,
                             float* __restrict__ gR, float* __restrict__ gt) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int p = 0; p < P; ++p) {
        const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float gi = g[(s * P + p) * 3 + i];
            a[3 * i] += gi * x; a[3 * i + 1] += gi * y; a[3 * i + 2] += gi * z;
            b[i] += gi;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) gR[s * 9 + k] = a[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) gt[s * 3 + k] = b[k];
}

// Frame.get_quaternion frame.py:87-114 (trace method, XYZW) and Frame.get_euler frame.py:120-121
__global__ void __launch_bounds__(256)
k_frame_quat_euler(const float* __restrict__ R, int64_t n, int stride, int pitch, float* __restrict__ quat_xyzw,
                   float* __restrict__ euler) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float* m = R + s * stride;
    float r[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = m[i * pitch + j];
    if (quat_xyzw) {
        float q[4];
        frame_quat_wxyz(r, q);
        *reinterpret_cast<float4*>(quat_xyzw + s * 4) = make_float4(q[1], q[2], q[3], q[0]);
    }
    if (euler) {
        euler[s * 3] = atan2f(r[7], r[8]);
        euler[s * 3 + 1] = asinf(-r[6]);
        euler[s * 3 + 2] = atan2f(r[3], r[0]);
    }
}

// Reverse mode of k_frame_quat_euler w.r.t. the rotation: gR [n,9] = d<gquat, quat> / dR + d<geuler, euler> / dR.
// Quaternion: the reference scales the trace-method vector by `0.5 / math.sqrt(tn * M[3][3])` -- a PYTHON float, so autograd
// sees the scale as a constant (frame.py:112); the same is done here: gR = sc * (d v / d R)^T g, v the unscaled vector.
// Euler (frame.py:120-121): atan2 / asin derivatives as torch gives them.
__global__ void __launch_bounds__(256)
k_frame_quat_euler_bwd(const float* __restrict__ R, int64_t n, int stride, int pitch, const float* __restrict__ gquat_xyzw,
                       const float* __restrict__ geuler, float* __restrict__ gR) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float* m = R + s * stride;
    float r[9], g[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r[3 * i + j] = m[i * pitch + j];
    if (gquat_xyzw) {
        const float4 gq = reinterpret_cast<const float4*>(gquat_xyzw)[s];
        float t = r[0] + r[4] + r[8] + 1.0f;
        // (a, b, c): adjoints of the three difference / sum terms, gd: adjoint of the diagonal term t
        if (t > 1.0f) {
            const float sc = 0.5f / sqrtf(t);
            const float gw = sc * gq.w, gx = sc * gq.x, gy = sc * gq.y, gz = sc * gq.z;
            g[0] += gw; g[4] += gw; g[8] += gw;
            g[3] += gz; g[1] -= gz; g[2] += gy; g[6] -= gy; g[7] += gx; g[5] -= gx;
        } else {
            int i = 0;
            if (r[4] > r[0]) i = 1;
            if (r[8] > (i == 0 ? r[0] : r[4])) i = 2;
            if (i == 0) {
                t = r[0] - (r[4] + r[8]) + 1.0f;
                const float sc = 0.5f / sqrtf(t);
                const float gw = sc * gq.w, gx = sc * gq.x, gy = sc * gq.y, gz = sc * gq.z;
                g[0] += gx; g[4] -= gx; g[8] -= gx;
                g[1] += gy; g[3] += gy; g[6] += gz; g[2] += gz; g[7] += gw; g[5] -= gw;
            } else if (i == 1) {
                t = r[4] - (r[8] + r[0]) + 1.0f;
                const float sc = 0.5f / sqrtf(t);
                const float gw = sc * gq.w, gx = sc * gq.x, gy = sc * gq.y, gz = sc * gq.z;
                g[4] += gy; g[8] -= gy; g[0] -= gy;
                g[5] += gz; g[7] += gz; g[1] += gx; g[3] += gx; g[2] += gw; g[6] -= gw;
            } else {
                t = r[8] - (r[0] + r[4]) + 1.0f;
                const float sc = 0.5f / sqrtf(t);
                const float gw = sc * gq.w, gx = sc * gq.x, gy = sc * gq.y, gz = sc * gq.z;
                g[8] += gz; g[0] -= gz; g[4] -= gz;
                g[6] += gx; g[2] += gx; g[5] += gy; g[7] += gy; g[3] += gw; g[1] -= gw;
            }
        }
    }
    if (geuler) {
        const float g0 = geuler[s * 3], g1 = geuler[s * 3 + 1], g2 = geuler[s * 3 + 2];
        const float d0 = r[7] * r[7] + r[8] * r[8], d2 = r[3] * r[3] + r[0] * r[0];
        g[7] += g0 * r[8] / d0; g[8] -= g0 * r[7] / d0;            // atan2(r21, r22)
        g[6] -= g1 / sqrtf(1.0f - r[6] * r[6]);                    // asin(-r20)
        g[3] += g2 * r[0] / d2; g[0] -= g2 * r[3] / d2;            // atan2(r10, r00)
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) gR[s * 9 + k] = g[k];
}

// x_rot / y_rot / z_rot spatial_vector.py:8-47 (axis 0 / 1 / 2) and q_to_rotation_matrix quaternion.py:102-120 (axis 3, input
// wxyz [n,4]): -> R [n,9].  gR != nullptr: reverse mode, gin[s] = <gR_s, dR/d in> (one angle, or the four quaternion components).
__global__ void __launch_bounds__(256)
k_rotation_from(int axis, const float* __restrict__ in, int64_t n, float* __restrict__ R, const float* __restrict__ gR,
                float* __restrict__ gin) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float r[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (axis == 3) {
        const float4 q = reinterpret_cast<const float4*>(in)[s];
        const float w = q.x, x = q.y, y = q.z, z = q.w;
        const float dc = 2.0f / (((w * w + x * x) + y * y) + z * z);        // torch sums the squares left to right
        r[0] = 1.0f - dc * (y * y + z * z); r[1] = dc * (x * y - z * w); r[2] = dc * (x * z + y * w);
        r[3] = dc * (x * y + z * w); r[4] = 1.0f - dc * (x * x + z * z); r[5] = dc * (y * z - x * w);
        r[6] = dc * (x * z - y * w); r[7] = dc * (y * z + x * w); r[8] = 1.0f - dc * (x * x + y * y);
        if (gR) {
            // R = I + dc M(q), M quadratic in q, dc = 2 / |q|^2:  d/dq_k <g, R> = dc <g, dM/dq_k> - dc^2 q_k <g, M>
            const float* g = gR + s * 9;
            const float gM = -g[0] * (y * y + z * z) + g[1] * (x * y - z * w) + g[2] * (x * z + y * w) +
                             g[3] * (x * y + z * w) - g[4] * (x * x + z * z) + g[5] * (y * z - x * w) +
                             g[6] * (x * z - y * w) + g[7] * (y * z + x * w) - g[8] * (x * x + y * y);
            const float dw = z * (g[3] - g[1]) + y * (g[2] - g[6]) + x * (g[7] - g[5]);
            const float dx = y * (g[1] + g[3]) + z * (g[2] + g[6]) + w * (g[7] - g[5]) - 2.0f * x * (g[4] + g[8]);
            const float dy = x * (g[1] + g[3]) + z * (g[5] + g[7]) + w * (g[2] - g[6]) - 2.0f * y * (g[0] + g[8]);
            const float dz = x * (g[2] + g[6]) + y * (g[5] + g[7]) + w * (g[3] - g[1]) - 2.0f * z * (g[0] + g[4]);
            const float k2 = dc * dc * gM;
            reinterpret_cast<float4*>(gin)[s] = make_float4(dc * dw - k2 * w, dc * dx - k2 * x, dc * dy - k2 * y, dc * dz - k2 * z);
        }
    } else {
        const float a = in[s];
        const float c = cosf(a), sn = sinf(a);
        // (i, j): the two axes the rotation mixes, in the sign convention of the reference
        const int i = axis == 0 ? 1 : 0, j = axis == 2 ? 1 : 2;
        const float sg = axis == 1 ? -1.0f : 1.0f;                           // y_rot has +sin above the diagonal
        r[4 * axis] = 1.0f;
        r[3 * i + i] = c; r[3 * j + j] = c; r[3 * i + j] = -sg * sn; r[3 * j + i] = sg * sn;
        if (gR) {
            const float* g = gR + s * 9;
            gin[s] = -sn * (g[3 * i + i] + g[3 * j + j]) + c * sg * (g[3 * j + i] - g[3 * i + j]);
        }
    }
    if (R) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[s * 9 + k] = r[k];
    }
}

// voxel grid of a cost model: (sdf [n0,n1,n2], grad [n0,n1,n2,3]) -> the tiled record table (grid_record), record = (gx, gy, gz, sdf);
// the dimensions are padded to whole 4 x 4 x 4 bricks (the padding is never addressed: cell indices are clamped to the grid)
__global__ void __launch_bounds__(256)
k_grid_pack(const float* __restrict__ sdf, const float* __restrict__ grad, int n0, int n1, int n2, int nb1, int nb2, int64_t n_rec,
            float4* __restrict__ cells) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const int64_t brick = r >> 6;
    const int in = (int)(r & 63);
    const int bk = (int)(brick % nb2), bj = (int)((brick / nb2) % nb1), bi = (int)(brick / ((int64_t)nb1 * nb2));
    const int i = 4 * bi + ((in >> 4) & 2) + ((in >> 2) & 1), j = 4 * bj + ((in >> 3) & 2) + ((in >> 1) & 1), k = 4 * bk + ((in >> 2) & 2) + (in & 1);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (i < n0 && j < n1 && k < n2) {
        const int64_t c = ((int64_t)i * n1 + j) * n2 + k;
        v = make_float4(grad[3 * c], grad[3 * c + 1], grad[3 * c + 2], sdf[c]);
    }
    cells[r] = v;
}

// GridMapSDF.precompute_sdf grid_map_sdf.py:34-63 (analytic objects only) and
// ObjectField.compute_signed_distance on arbitrary points
__global__ void __launch_bounds__(256)
k_grid_precompute(DevCostHdr C, int nx, int ny, int nz, float lo0, float lo1, float lo2, float hi0, float hi1, float hi2,
                  float* __restrict__ sdf, float* __restrict__ grad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)nx * ny * nz;
    if (idx >= total) return;
    const int iz = (int)(idx % nz), iy = (int)((idx / nz) % ny), ix = (int)(idx / ((int64_t)nz * ny));
    const int id[3] = {ix, iy, iz}, dims[3] = {nx, ny, nz};
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    float x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // torch.linspace: start + i*step for the first half, end - (steps-1-i)*step after
        const float step = dims[k] > 1 ? (hi[k] - lo[k]) / (float)(dims[k] - 1) : 0.0f;
        x[k] = id[k] < dims[k] / 2 ? lo[k] + step * (float)id[k] : hi[k] - step * (float)(dims[k] - 1 - id[k]);
    }
    float best = 0.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
    bool first = true;
    for (int o = 0; o < C.n_objects; ++o) {
        if (cptr(C.objects)[o].is_grid) continue;
        float gx, gy, gz;
        const float v = object_sdf<true>(C, o, x[0], x[1], x[2], gx, gy, gz);
        const bool take = first || v < best;
        best = take ? v : best; bx = take ? gx : bx; by = take ? gy : by; bz = take ? gz : bz;
        first = false;
    }
    sdf[idx] = best;
    grad[3 * idx] = bx; grad[3 * idx + 1] = by; grad[3 * idx + 2] = bz;
}

__global__ void __launch_bounds__(256)
k_sdf_points(DevCostHdr C, const float* __restrict__ pts, int64_t n, float* __restrict__ sdf, float* __restrict__ grad) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float x = pts[3 * s], y = pts[3 * s + 1], z = pts[3 * s + 2];
    for (int o = 0; o < C.n_objects; ++o) {
        float gx, gy, gz;
        sdf[s * C.n_objects + o] = object_sdf<true>(C, o, x, y, z, gx, gy, gz);
        if (grad) {
            float* g = grad + (s * C.n_objects + o) * 3;
            g[0] = gx; g[1] = gy; g[2] = gz;
        }
    }
}

// interpolate_traj_via_points (trajectory/utils.py:37-50): between consecutive via points i, i+1 emit
// x_i * alpha_a + x_{i+1} * (1 - alpha_a) for the n interior alphas (the via points themselves are not emitted).
// x [T, H, D] -> out [T, (H-1)*n, D]; alpha, beta = 1 - alpha: DEVICE [n] (host computes torch.linspace).
// Two roundings per product and one for the sum, like the reference's `a * alpha + b * (1 - alpha)`.
// element-per-thread form: any trajectory size (used when a trajectory's way points do not fit the LDS of a workgroup)
__global__ void __launch_bounds__(256)
k_interpolate_via_points_flat(const float* __restrict__ x, int64_t T, int H, int D, int n_interp,
                              const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ out) {
    const int64_t total = T * (int64_t)(H - 1) * n_interp * D;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int d = (int)(idx % D);
    const int64_t r = idx / D;
    const int a = (int)(r % n_interp);
    const int64_t s = r / n_interp;
    const int i = (int)(s % (H - 1));
    const int64_t t = s / (H - 1);
    const float x0 = x[(t * H + i) * D + d], x1 = x[(t * H + i + 1) * D + d];
    out[idx] = __fadd_rn(__fmul_rn(x0, alpha[a]), __fmul_rn(x1, beta[a]));
}

// One workgroup per trajectory: its H x D way points are staged in LDS once (coalesced), then thread r produces output row
// r = i * n + a (D consecutive floats) -- one 32-bit division per row instead of four 64-bit ones per element.
__global__ void __launch_bounds__(256)
k_interpolate_via_points(const float* __restrict__ x, int64_t T, int H, int D, int n_interp,
                         const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float xs[];           // [H * D] way points, then alpha[n], beta[n]
    const int64_t t = blockIdx.x;
    const int hd = H * D;
    const float* xt = x + t * hd;
    for (int k = threadIdx.x; k < hd; k += blockDim.x) xs[k] = xt[k];
    float* al = xs + hd;
    float* be = al + n_interp;
    for (int k = threadIdx.x; k < n_interp; k += blockDim.x) { al[k] = alpha[k]; be[k] = beta[k]; }
    __syncthreads();
    const int rows = (H - 1) * n_interp;
    float* ot = out + t * (int64_t)rows * D;
    // one output ELEMENT per thread and trip (round 6; before: one row of D floats per thread, every store instruction a 4-byte-in-4D
    // stride pattern): consecutive threads write consecutive floats.  e = r D + d, r = i n + a: two small divisions by wave-uniform
    // divisors (reciprocal + fix-up, exact for e < 2^22; larger trajectories take the integer divisions)
    const int total = rows * D;
    const float inv_D = 1.0f / (float)D, inv_n = 1.0f / (float)n_interp;
    const bool small = total < (1 << 22);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = small ? trk_div_small(e, D, inv_D) : e / D, d = e - r * D;
        const int i = small ? trk_div_small(r, n_interp, inv_n) : r / n_interp, a = r - i * n_interp;
        const float* p0 = xs + i * D + d;
        ot[e] = __fadd_rn(__fmul_rn(p0[0], al[a]), __fmul_rn(p0[D], be[a]));
    }
}

// ============================================================================================
// Trajectory validation, device side (PlanningTask.get_trajs_collision_and_free tasks.py:253-308): the per-way-point collision
// bytes are folded into per-trajectory flags, the trajectories are partitioned in the reference's order, and the two groups are
// gathered -- all queued back to back; the host reads three counters once.
// ============================================================================================
// one wavefront per trajectory: bit 0 = some interpolated way point is in collision (tasks.py:255-256), bit 1 = some joint
// position of the ORIGINAL way points lies outside [q_min, q_max] (tasks.py:270-273; a NaN counts as outside, as in torch).
// The loads of a trajectory are issued eight at a time before any is tested: one load per loop round made the kernel a chain
// of ~12 memory round trips (7.6 us at 4096 x 64 for 9 MB).
#define TRK_FLAGS_BATCH 8
__global__ void __launch_bounds__(256)
k_traj_flags(const uint8_t* __restrict__ wp, int Hi, const float* __restrict__ x, int H, int S, int D,
             const float* __restrict__ qmin, const float* __restrict__ qmax, int64_t T, uint8_t* __restrict__ flags) {
    const int lane = threadIdx.x & (TRK_WAVE - 1);
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x / TRK_WAVE) + threadIdx.x / TRK_WAVE;
    if (t >= T) return;
    bool coll = false, outside = false;
    const uint8_t* w = wp + t * Hi;
    const float* xt = x + t * (int64_t)H * S;
    const int nx = H * S;
    for (int k0 = lane; k0 < Hi; k0 += TRK_WAVE * TRK_FLAGS_BATCH) {
        uint8_t v[TRK_FLAGS_BATCH];
#pragma unroll
        for (int j = 0; j < TRK_FLAGS_BATCH; ++j) { const int k = k0 + TRK_WAVE * j; v[j] = k < Hi ? w[k] : 0; }
#pragma unroll
        for (int j = 0; j < TRK_FLAGS_BATCH; ++j) coll |= v[j] != 0;
    }
    for (int k0 = lane; k0 < nx; k0 += TRK_WAVE * TRK_FLAGS_BATCH) {       // contiguous read; columns >= D are not positions
        float v[TRK_FLAGS_BATCH];
#pragma unroll
        for (int j = 0; j < TRK_FLAGS_BATCH; ++j) { const int k = k0 + TRK_WAVE * j; v[j] = k < nx ? xt[k] : 0.0f; }
#pragma unroll
        for (int j = 0; j < TRK_FLAGS_BATCH; ++j) {
            const int k = k0 + TRK_WAVE * j, d = k % S;
            if (k < nx && d < D) outside |= !(v[j] >= qmin[d] && v[j] <= qmax[d]);
        }
    }
    const bool any_c = __ballot(coll) != 0, any_o = __ballot(outside) != 0;
    if (lane == 0) flags[t] = (any_c ? 1 : 0) | (any_o ? 2 : 0);
}

// One workgroup; thread i owns the contiguous block of trajectories [i * chunk, (i + 1) * chunk), so one exclusive scan of the
// per-thread counts gives every thread its write positions and the order inside a group is the trajectory order.  A stable
// three-way partition into ONE index list:
//   rows [0, n_free)                     collision free AND inside the joint limits         (tasks.py:274, 282)
//   rows [n_free, n_free + n_coll)       colliding                                          (tasks.py:256)
//   rows [.., .. + n_out)                collision free but outside the limits              (tasks.py:278-281)
// as int64 rows [t] (inner == 0) or [t / inner, t % inner] (a 4-D batch), like torch.argwhere; counts = {free, colliding, outside}.
// partial != nullptr (round 6): the flags are first assembled from the per-wavefront bytes of the via-point launch
// (trk_rollout_collision_via_flags; SpecArgs::via_partial: one short row per trajectory).
#define TRK_PARTITION_LDS_FLAGS 32768
__global__ void __launch_bounds__(1024)
k_traj_partition(uint8_t* __restrict__ flags, const uint8_t* __restrict__ partial, int64_t hi, int n_slots, int64_t T, int64_t inner,
                 int64_t* __restrict__ idx, int32_t* __restrict__ counts, int32_t* __restrict__ counts_host, int32_t ticket) {
    __shared__ int wsum[3][16];
    __shared__ uint8_t sflags[TRK_PARTITION_LDS_FLAGS];          // the assembled flags of up to this many trajectories stay on the CU
    const int tid = threadIdx.x, lane = tid & (TRK_WAVE - 1), wave = tid / TRK_WAVE;
    const int64_t chunk = (T + 1023) / 1024;
    const int64_t t0 = min(T, (int64_t)tid * chunk), t1 = min(T, t0 + chunk);
    const bool in_lds = partial != nullptr && T <= TRK_PARTITION_LDS_FLAGS;
    if (partial) {
        // phase 0: the flags, assembled cooperatively (consecutive threads take consecutive trajectories): trajectory t is covered by the
        // wavefronts floor(t hi / 64) .. floor(((t + 1) hi - 1) / 64), whose bytes are the first entries of its row of K = hi / 64 + 2
        const int64_t K = hi / TRK_WAVE + 2;
        for (int64_t t = tid; t < T; t += 1024) {
            const int nj = (int)(((t + 1) * hi - 1) / TRK_WAVE - t * hi / TRK_WAVE) + 1;
            const uint8_t* row = partial + t * K;
            int f = 0;
            if (K <= 8) {                       // the common case (up to 384 samples per trajectory): the whole row in flight at once
                uint8_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = row[k < nj ? k : 0];
#pragma unroll
                for (int k = 0; k < 8; ++k) f |= v[k];
            } else {
                for (int k = 0; k < nj; ++k) f |= row[k];
            }
            flags[t] = (uint8_t)f;
            if (in_lds) sflags[t] = (uint8_t)f;
        }
        __threadfence_block();
        __syncthreads();
    }
    int c[3] = {0, 0, 0};
    for (int64_t t = t0; t < t1; ++t) {
        const int f = in_lds ? sflags[t] : flags[t];
        c[0] += f == 0; c[1] += f & 1; c[2] += f == 2;
    }
    int incl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {               // inclusive scan inside the wavefront, then across the 16 wavefronts
        int v = c[k];
        for (int o = 1; o < TRK_WAVE; o <<= 1) {
            const int u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        incl[k] = v;
        if (lane == TRK_WAVE - 1) wsum[k][wave] = v;
    }
    __syncthreads();
    int off[3], tot[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int before = 0, all = 0;
        for (int w2 = 0; w2 < 16; ++w2) { const int s = wsum[k][w2]; all += s; before += w2 < wave ? s : 0; }
        off[k] = before + incl[k] - c[k];
        tot[k] = all;
    }
    off[1] += tot[0]; off[2] += tot[0] + tot[1];
    const int cols = inner > 0 ? 2 : 1;
    for (int64_t t = t0; t < t1; ++t) {
        const int f = in_lds ? sflags[t] : flags[t];
        int64_t* dst = idx + (int64_t)(f == 0 ? off[0]++ : ((f & 1) ? off[1]++ : off[2]++)) * cols;
        if (inner > 0) { dst[0] = t / inner; dst[1] = t % inner; }
        else dst[0] = t;
    }
    if (tid == 0) {
        counts[0] = tot[0]; counts[1] = tot[1]; counts[2] = tot[2]; counts[3] = ticket;
        if (counts_host) {
            // pinned host memory, written straight from the kernel: the three counters, then -- with system-scope release
            // semantics -- the caller's ticket, which the host polls for (no copy call, no event)
            counts_host[0] = tot[0]; counts_host[1] = tot[1]; counts_host[2] = tot[2];
            __hip_atomic_store(&counts_host[3], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// out[r] = x[idx[r]] for every row of the partitioned list: the free trajectories first, then the others (one workgroup per row)
__global__ void __launch_bounds__(256)
k_traj_gather(const float* __restrict__ x, int row, int cols, const int64_t* __restrict__ idx, int64_t inner, float* __restrict__ out) {
    const int64_t r = blockIdx.x;
    const int64_t* ix = idx + r * cols;
    const int64_t t = inner > 0 ? ix[0] * inner + ix[1] : ix[0];
    const float* src = x + t * row;
    float* dst = out + r * row;
    if ((row & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int k = threadIdx.x; k < row / 4; k += blockDim.x) d4[k] = s4[k];
    } else {
        for (int k = threadIdx.x; k < row; k += blockDim.x) dst[k] = src[k];
    }
}

// ============================================================================================
// Gauss-Newton normal equations of one link's geometric Jacobian (robot_tree.py:238-246): per sample J = [lin_jac; ang_jac]
// (6 x D) -> JtJ (D x D) = J^T J and Jtr (D) = J^T r.  BUILD-DEFINED (the reference stops at the Jacobian; BASELINE's north star
// names "the small dense Jacobian x Jacobian blocks" as the place for MFMA "where it is a real contraction").
// Two kernels with the same I/O skeleton (Jacobians in through one coalesced LDS tile, results out through another):
//   k_jtj<false>  VALU: one lane per sample, the D (D + 1) / 2 distinct entries x 6 FMAs, mirrored on the way out;
//   k_jtj<true>   MFMA: v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 outer products per instruction; a sample's 8x8-padded
//                 JtJ is 4 such blocks, so one instruction advances 4 samples by one row k of J, 6 instructions finish them.
// fp32 MFMA runs at the fp32 VECTOR rate on gfx950 (64 FLOP/clk/SIMD, MI355X_MICROARCH.md), computes the padded and the mirrored
// entries as well, and the op moves 4 (12 D + 6 + D^2 + D) bytes per sample for ~12 D^2 flops: it is HBM-bound either way
// (profiles/r03_bench_jtj.txt has both kernels side by side).
// ============================================================================================
typedef float trk_v4 __attribute__((ext_vector_type(4)));
#define TRK_JTJ_WAVES 1      // one wavefront per workgroup: the two tiles are 20 - 75 KB per wavefront, LDS decides the occupancy
// DT > 0 (round 6): the DOF count as a compile-time constant for the common arms (6, 7) -- a lane takes its 6 x D Jacobian out of the
// tile ONCE (42 LDS reads instead of 12 per entry = 336), the D (D + 1) / 2 + D results stay in registers, and the output tile REUSES the
// input tile's LDS (14.6 instead of 25.6 KB per wavefront at 7 DOF: 10 instead of 6 wavefronts per CU).  Same arithmetic, same order.
template <bool MFMA, int DT = 0>
__global__ void __launch_bounds__(TRK_JTJ_WAVES * TRK_WAVE)
k_jtj(const float* __restrict__ lin, const float* __restrict__ ang, const float* __restrict__ r6, int64_t n, int D_arg,
      float* __restrict__ JtJ, float* __restrict__ Jtr, const float* __restrict__ damping, int damping_stride, float* __restrict__ dq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(!(MFMA && DT > 0), "the compile-time-D path is the per-lane kernel");
    const int D = DT > 0 ? DT : D_arg;
    const int lane = threadIdx.x & (TRK_WAVE - 1), wave = threadIdx.x / TRK_WAVE;
    const int64_t base = ((int64_t)blockIdx.x * TRK_JTJ_WAVES + wave) * TRK_WAVE;
    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, n - base));
    const int DP = MFMA ? 8 : D;                           // padded row length of a Jacobian row in LDS
    const int js = (6 * DP) | 1;                            // per-sample stride of the Jacobian tile (odd: conflict-free lanes)
    const int os = (D * D + D) | 1;                         // per-sample stride of the output tile [JtJ | Jtr]
    float* jt = smem + wave * TRK_WAVE * (DT > 0 ? (js > os ? js : os) : (js + os));      // [64][js]: a wavefront works on its own tiles, no workgroup barrier
    float* ot = DT > 0 ? jt : jt + TRK_WAVE * js;           // [64][os]; DT > 0: the same LDS, written after every lane has taken its Jacobian out
    if (MFMA) for (int k = lane; k < TRK_WAVE * js; k += TRK_WAVE) jt[k] = 0.0f;      // zero padding columns D..7
    // J rows 0..2 = lin_jac[s, :, :], 3..5 = ang_jac[s, :, :]: each array is one contiguous run of rows * 3D floats per wavefront
    const int w3 = 3 * D, dd = D * D;
    const float inv_w3 = 1.0f / (float)w3, inv_D = 1.0f / (float)D, inv_dd = 1.0f / (float)dd;
    const float* lsrc = lin + base * w3;
    const float* asrc = ang + base * w3;
    auto put = [&](int k, float lv, float av) {            // element k of the wavefront's run -> its place in the tile
        const int s = trk_div_small(k, w3, inv_w3), e = k - s * w3, row = trk_div_small(e, D, inv_D), d = e - row * D;
        jt[s * js + row * DP + d] = lv;
        jt[s * js + (3 + row) * DP + d] = av;
    };
    if (rows == TRK_WAVE && ((reinterpret_cast<uintptr_t>(lsrc) | reinterpret_cast<uintptr_t>(asrc)) & 15) == 0) {
        // 64 * 3D floats = 48 D float4 per array: all of a lane's 16-byte loads are issued before the first LDS write
        const int nv = 48 * D;                              // float4 count (64 * 3D / 4)
        for (int v0 = lane; v0 < nv; v0 += TRK_WAVE * 4) {
            float4 a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = v0 + TRK_WAVE * j;
                if (v < nv) { a[j] = reinterpret_cast<const float4*>(lsrc)[v]; b[j] = reinterpret_cast<const float4*>(asrc)[v]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = v0 + TRK_WAVE * j;
                if (v < nv) { put(4 * v, a[j].x, b[j].x); put(4 * v + 1, a[j].y, b[j].y); put(4 * v + 2, a[j].z, b[j].z); put(4 * v + 3, a[j].w, b[j].w); }
            }
        }
    } else {
        for (int k = lane; k < rows * w3; k += TRK_WAVE) put(k, lsrc[k], asrc[k]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* mine = jt + lane * js;
    float* out = ot + lane * os;
    if constexpr (DT > 0) {
        float Jr[6][DT], rv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int i = 0; i < DT; ++i) Jr[k][i] = mine[k * DT + i];
#pragma unroll
        for (int k = 0; k < 6; ++k) rv[k] = ((Jtr || dq) && r6 && lane < rows) ? r6[(base + lane) * 6 + k] : 0.0f;
        float A[DT][DT], bt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i) {
#pragma unroll
            for (int j = i; j < DT; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) acc = fmaf(Jr[k][i], Jr[k][j], acc);
                A[i][j] = acc;
            }
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = fmaf(Jr[k][i], rv[k], acc);
            bt[i] = acc;
        }
        // every lane has read its Jacobian: the tile may now be overwritten with the results
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < DT; ++i) {
#pragma unroll
            for (int j = i; j < DT; ++j) { out[i * DT + j] = A[i][j]; out[j * DT + i] = A[i][j]; }
            if (Jtr || dq) out[DT * DT + i] = bt[i];
        }
    } else if (!MFMA) {
        for (int i = 0; i < D; ++i)
            for (int j = i; j < D; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) acc = fmaf(mine[k * DP + i], mine[k * DP + j], acc);
                out[i * D + j] = acc; out[j * D + i] = acc;
            }
    } else {
        // lane l of the instruction: block b = l / 4 (16 blocks), element e = l % 4.  Block b belongs to sample 4 g + b / 4 of
        // the wavefront and is tile (bi, bj) = ((b % 4) / 2, b % 2) of that sample's 8 x 8 matrix.  Operand A carries
        // J[k][4 bi + e], operand B J[k][4 bj + e]; accumulator register r then holds JtJ[4 bi + r][4 bj + e].
        const int b = lane >> 2, e = lane & 3, bi = (b & 3) >> 1, bj = b & 1;
        for (int g = 0; g < TRK_WAVE / 4; ++g) {
            const float* src = jt + (4 * g + (b >> 2)) * js;
            trk_v4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 6; ++k)
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(src[k * 8 + 4 * bi + e], src[k * 8 + 4 * bj + e], acc, 0, 0, 0);
            float* dst = ot + (4 * g + (b >> 2)) * os;
            const int col = 4 * bj + e;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * bi + r;
                if (row < D && col < D) dst[row * D + col] = acc[r];
            }
        }
    }
    if (DT == 0 && (Jtr || dq)) {
        float rv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) rv[k] = (r6 && lane < rows) ? r6[(base + lane) * 6 + k] : 0.0f;
        for (int i = 0; i < D; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = fmaf(mine[k * DP + i], rv[k], acc);
            out[D * D + i] = acc;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* jdst = JtJ + base * dd;
    auto get = [&](int k) { const int s = trk_div_small(k, dd, inv_dd); return ot[s * os + (k - s * dd)]; };
    if (rows == TRK_WAVE && (reinterpret_cast<uintptr_t>(jdst) & 15) == 0) {
        for (int v = lane; v < 16 * dd; v += TRK_WAVE)     // 64 * D^2 floats = 16 D^2 float4
            reinterpret_cast<float4*>(jdst)[v] = make_float4(get(4 * v), get(4 * v + 1), get(4 * v + 2), get(4 * v + 3));
    } else {
        for (int k = lane; k < rows * dd; k += TRK_WAVE) jdst[k] = get(k);
    }
    if (Jtr) for (int k = lane; k < rows * D; k += TRK_WAVE) { const int s = trk_div_small(k, D, inv_D); Jtr[base * D + k] = ot[s * os + dd + (k - s * D)]; }
    if (dq) {
        // damped Gauss-Newton / Levenberg-Marquardt step per sample: (JtJ + lambda I) dq = Jtr by an in-place Cholesky factorisation
        // in the lane's own row of the output tile (already copied out above), then the two triangular solves
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float lam = (damping && lane < rows) ? damping[(base + lane) * damping_stride] : 0.0f;
        float* Am = out;                                    // [D][D] row-major, lower triangle becomes L
        float* bv = out + dd;                               // right-hand side -> y -> x
        for (int j = 0; j < D; ++j) {
            float sdiag = Am[j * D + j] + lam;
            for (int k = 0; k < j; ++k) sdiag = fmaf(-Am[j * D + k], Am[j * D + k], sdiag);
            const float ljj = sqrtf(fmaxf(sdiag, 1e-20f));
            const float inv = 1.0f / ljj;
            Am[j * D + j] = ljj;
            for (int i = j + 1; i < D; ++i) {
                float v = Am[i * D + j];
                for (int k = 0; k < j; ++k) v = fmaf(-Am[i * D + k], Am[j * D + k], v);
                Am[i * D + j] = v * inv;
            }
        }
        for (int i = 0; i < D; ++i) {                       // L y = b
            float v = bv[i];
            for (int k = 0; k < i; ++k) v = fmaf(-Am[i * D + k], bv[k], v);
            bv[i] = v / Am[i * D + i];
        }
        for (int i = D - 1; i >= 0; --i) {                  // L^T x = y
            float v = bv[i];
            for (int k = i + 1; k < D; ++k) v = fmaf(-Am[k * D + i], bv[k], v);
            bv[i] = v / Am[i * D + i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = lane; k < rows * D; k += TRK_WAVE) { const int s = trk_div_small(k, D, inv_D); dq[base * D + k] = ot[s * os + dd + (k - s * D)]; }
    }
}

// out[n, :] = g[n, :] * s[n]: the chain rule of the fused rollout under autograd -- its saved d cost[n] / d q[n, :] times the
// upstream gradient of cost[n] (tasks.py:135-137 followed by .backward()).  fp32 or fp16 rows, fp32 scale; one thread per element.
template <class IO>
__global__ void __launch_bounds__(256)
k_scale_rows(const IO* __restrict__ g, const float* __restrict__ sc, int sc_stride, int64_t n, int D, IO* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * D) return;
    out[idx] = (IO)((float)g[idx] * (sc_stride ? sc[idx / D] : sc[0]));
}

// interpolate_points_v1 distance_fields.py:66-69 (F.interpolate linear, align_corners=True, along the link axis) with the index /
// weight table unrolled on the host: x [N, L, C] -> out [N, K, C], out[n, k] = w[2k] x[n, src[2k]] + w[2k+1] x[n, src[2k+1]].
// One thread per output element: consecutive threads write consecutive floats.
__global__ void __launch_bounds__(256)
k_interpolate_columns(const float* __restrict__ x, int64_t n, int L, int C, int K, const int32_t* __restrict__ src,
                      const float* __restrict__ w, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * K * C) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int k = (int)(r % K);
    const float* xs = x + (r / K) * L * C + c;
    out[idx] = __fadd_rn(__fmul_rn(w[2 * k], xs[src[2 * k] * C]), __fmul_rn(w[2 * k + 1], xs[src[2 * k + 1] * C]));
}
// its reverse mode: g [N, K, C] -> gx [N, L, C]; one thread per input element gathers from the output points that read it
__global__ void __launch_bounds__(256)
k_interpolate_columns_bwd(const float* __restrict__ g, int64_t n, int L, int C, int K, const int32_t* __restrict__ src,
                          const float* __restrict__ w, float* __restrict__ gx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * L * C) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int l = (int)(r % L);
    const float* gs = g + (r / L) * K * C + c;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        if (src[2 * k] == l) acc = fmaf(w[2 * k], gs[k * C], acc);
        if (src[2 * k + 1] == l) acc = fmaf(w[2 * k + 1], gs[k * C], acc);
    }
    gx[idx] = acc;
}

// ============================================================================================
// Constant-velocity GP prior over a trajectory (GPMP-style smoothness term; BUILD-DEFINED: the reference has only
// finite differences, SURVEY.md 8a "items the north star names but the reference does not contain").
// Per DOF the state is x_t = (p_t, v_t);  e_t = Phi x_t - x_{t+1} = (p_t + dt v_t - p_{t+1}, v_t - v_{t+1});
// Q^-1 = sigma^-2 [[12/dt^3, -6/dt^2], [-6/dt^2, 4/dt]];  cost_b = w * 1/2 sum_t sum_d e^T Q^-1 e.
// With r_t = Q^-1 e_t:  d/dx_t = Phi^T r_t - r_{t-1} = (rp_t - rp_{t-1}, dt rp_t + rv_t - rv_{t-1}).
// The DOFs do not couple (Q^-1 is 2x2 (x) I_D), so there is no matrix contraction to hand to MFMA: this is a
// streaming kernel.  One workgroup per trajectory; element (t, d) -> thread, fully coalesced; T = float or _Float16 I/O
// with fp32 arithmetic and an fp32 cost.
// ============================================================================================
template <class T> struct GpVec;          // 4 consecutive elements <-> float4
template <> struct GpVec<float> {
    static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    // write-through like the fused kernel's outputs: plain stores leave the gradients dirty in the L2s and their write-back at
    // the kernel boundary stalls the next launch (measured on the rollout kernel: 12.97 -> 10.87 us)
    static __device__ __forceinline__ void store(float* p, const float4& v) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 x = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
    }
};
template <> struct GpVec<_Float16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float4 load(const _Float16* p) {
        const h4 h = *reinterpret_cast<const h4*>(p);
        return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
    }
    // gradients: saturating at the largest finite half (at sigma = 0.1, dt = 5/128 a residual of 0.02 rad is a gradient of 4e5;
    // the caller's grad_scale keeps the values in range, this keeps a misjudged scale from writing inf)
    static __device__ __forceinline__ _Float16 sat(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
    static __device__ __forceinline__ void store(_Float16* p, const float4& v) {
        const h4 h = {sat(v.x), sat(v.y), sat(v.z), sat(v.w)};
        asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(h) : "memory");
    }
};

// One workgroup per trajectory.  VEC: H*D is a multiple of 4 and the buffers are 16-byte aligned -> the trajectory's q
// and qd go to LDS with 4-element loads, each thread then owns 4 consecutive elements and writes them with one store.
// T: HBM-side type of q / qd, G: of gq / gqd.  The gradients are multiplied by `gs` (the caller's grad_scale) before they are
// stored (or added to what the buffers hold, which the caller scaled the same way); the cost is never scaled.
template <class T, class G, bool VEC>
__global__ void __launch_bounds__(256)
k_gp_prior(const T* __restrict__ q, const T* __restrict__ qd, int H, int D, float dt, float a, float b, float c, float w, float gs,
           float* __restrict__ cost, G* __restrict__ gq, G* __restrict__ gqd, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t base = (int64_t)blockIdx.x * H * D;
    const int total = H * D;
    float* ps = smem;
    float* vs = smem + ((total + 3) & ~3);
    float* part = vs + ((total + 3) & ~3);        // 4 floats: per-wavefront partial sums
    if (VEC) {
        for (int i = 4 * threadIdx.x; i < total; i += 4 * 256) {
            *reinterpret_cast<float4*>(ps + i) = GpVec<T>::load(q + base + i);
            *reinterpret_cast<float4*>(vs + i) = GpVec<T>::load(qd + base + i);
        }
    } else {
        for (int i = threadIdx.x; i < total; i += 256) { ps[i] = (float)q[base + i]; vs[i] = (float)qd[base + i]; }
    }
    __syncthreads();
    float acc = 0.0f;
    auto element = [&](int i, float& gp, float& gv) {
        const float p0 = ps[i], v0 = vs[i];
        gp = 0.0f; gv = 0.0f;
        if (i + D < total) {                       // t + 1 < H for t = i / D, without the division
            const float ep = fmaf(dt, v0, p0) - ps[i + D], ev = v0 - vs[i + D];
            const float rp = fmaf(a, ep, b * ev), rv = fmaf(b, ep, c * ev);
            acc = fmaf(0.5f, fmaf(ep, rp, ev * rv), acc);
            gp = rp; gv = fmaf(dt, rp, rv);
        }
        if (i >= D) {                              // t > 0
            const float pm = ps[i - D], vm = vs[i - D];
            const float ep = fmaf(dt, vm, pm) - p0, ev = vm - v0;
            gp -= fmaf(a, ep, b * ev); gv -= fmaf(b, ep, c * ev);
        }
        gp *= w * gs; gv *= w * gs;
    };
    if (VEC) {
        for (int i = 4 * threadIdx.x; i < total; i += 4 * 256) {
            float gp[4], gv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) element(i + k, gp[k], gv[k]);
            float4 o0 = make_float4(gp[0], gp[1], gp[2], gp[3]), o1 = make_float4(gv[0], gv[1], gv[2], gv[3]);
            if (accumulate) {
                const float4 c0 = GpVec<G>::load(gq + base + i), c1 = GpVec<G>::load(gqd + base + i);
                o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w;
                o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
            }
            GpVec<G>::store(gq + base + i, o0);
            GpVec<G>::store(gqd + base + i, o1);
        }
    } else {
        for (int i = threadIdx.x; i < total; i += 256) {
            float gp, gv;
            element(i, gp, gv);
            if (accumulate) { gp += (float)gq[base + i]; gv += (float)gqd[base + i]; }
            put_scaled(gq + base + i, gp); put_scaled(gqd + base + i, gv);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && cost) cost[blockIdx.x] = w * ((part[0] + part[1]) + (part[2] + part[3]));
}

// The two small kernels of the TWO-LAUNCH form of trk_rollout_gp_cost_grad (robots / cost models the fused generated kernel does not
// serve): the prior's factor between t and t + 1 is attributed to sample (b, t), cost[b, t] += w/2 e_t^T Q^-1 e_t, and the
// per-wavefront cost sums are re-formed from the finished costs (same association order as the fused kernels' DPP reduction is not
// needed: the sums are a convenience output, compared to rounding).
template <class T>
__global__ void __launch_bounds__(256)
k_gp_sample_cost(const T* __restrict__ q, const T* __restrict__ qd, int64_t n, int H, int D, float dt, float a, float b, float c, float w,
                 float* __restrict__ cost) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int t = (int)(s % H);
    if (t + 1 >= H) return;
    const T* p = q + s * D; const T* v = qd + s * D;
    float acc = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float p0 = (float)p[d], v0 = (float)v[d];
        const float ep = fmaf(dt, v0, p0) - (float)p[D + d], ev = v0 - (float)v[D + d];
        const float rp = fmaf(a, ep, b * ev), rv = fmaf(b, ep, c * ev);
        acc = fmaf(0.5f, fmaf(ep, rp, ev * rv), acc);
    }
    cost[s] += w * acc;
}
__global__ void __launch_bounds__(TRK_WAVE)
k_block_sums(const float* __restrict__ cost, int64_t n, float* __restrict__ sums) {
    const int64_t s = (int64_t)blockIdx.x * TRK_WAVE + threadIdx.x;
    const float tot = wave_sum(s < n ? cost[s] : 0.0f);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// ============================================================================================
// Trajectory plumbing of the reference (A17): zero-padded finite differences along the horizon
// (finite_difference_vector trajectory/utils.py:53-64, used by RobotBase.get_velocity / get_acceleration robot_base.py:151-166)
// and  sum_t || x[t+1] - x[t] ||  over selected state columns (compute_path_length / compute_smoothness
// trajectory/metrics.py:7-12, 27-35).  Streaming kernels; same operation order as the torch expressions.
// ============================================================================================
__global__ void __launch_bounds__(256)
k_finite_difference(const float* __restrict__ x, int64_t total, int H, int D, float dt, int method, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int t = (int)((idx / D) % H);
    float v = 0.0f;
    if (method == 0) { if (t + 1 < H) v = (x[idx + D] - x[idx]) / dt; }                       // forward:  diff[:-1] = (x[1:] - x[:-1]) / dt
    else if (method == 1) { if (t > 0) v = (x[idx] - x[idx - D]) / dt; }                       // backward: diff[1:]  = (x[1:] - x[:-1]) / dt
    else { if (t > 0 && t + 1 < H) v = (x[idx + D] - x[idx - D]) / (2.0f * dt); }             // central:  diff[1:-1] = (x[2:] - x[:-2]) / (2 dt)
    out[idx] = v;
}

// one workgroup per trajectory; x [B, H, S], columns [c0, c0 + D)
__global__ void __launch_bounds__(256)
k_traj_diff_norm_sum(const float* __restrict__ x, int H, int S, int c0, int D, float* __restrict__ out) {
    __shared__ float part[4];
    const float* tr = x + (int64_t)blockIdx.x * H * S + c0;
    float acc = 0.0f;
    for (int t = threadIdx.x; t + 1 < H; t += 256) {
        float n2 = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float df = tr[(int64_t)(t + 1) * S + d] - tr[(int64_t)t * S + d];
            n2 = fmaf(df, df, n2);
        }
        acc += sqrtf(n2);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// deterministic sum: one 256-thread workgroup, fixed strides, LDS tree
__global__ void __launch_bounds__(256)
k_reduce_sum(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float part[256];
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += x[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// the same for long vectors (n > TRK_REDUCE_WIDE_FROM): 1024 threads, 16-byte loads, eight in flight per thread -- one CU streams
// 1 MB in a few microseconds where the 256-thread loop above needs ~250.  Its own fixed association order (thread t adds the
// float4s t, t + 1024, ... left to right, lanes x + y + z + w, then the LDS tree); short vectors keep the order k_pack_sums mirrors.
#define TRK_REDUCE_WIDE_FROM 65536
__global__ void __launch_bounds__(1024)
k_reduce_sum_wide(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float part[1024];
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;     // unaligned views take the scalar tail loop for everything
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float acc = 0.0f;
    int64_t i = threadIdx.x;
    for (; i + 7 * 1024 < n4; i += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x4[i + k * 1024];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    for (; i < n4; i += 1024) { const float4 v = x4[i]; acc += (v.x + v.y) + (v.z + v.w); }
    for (int64_t j = 4 * n4 + threadIdx.x; j < n; j += 1024) acc += x[j];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// ============================================================================================
// What a batch-sharded planner exchanges: packed = [ sum cost | sum_b cost(b, h) (H) | sum_b gq(b, h, d) (H D) ]
// of one rank's evaluation, bit-reproducibly.  cost [B, H]; gq [B, H, D]; block_sums [nb] (trk_rollout_cost_grad).
// The C = H + H D columns are cut into tiles of 64 UNITS (a unit = V consecutive columns: V = 4 -- one 16-byte load, 8 bytes of an fp16
// gradient -- when H is a multiple of 4 and the arrays are aligned, else 1) and the B trajectories into `slices` row slices.
// k_pack_partial: a workgroup (512 threads = 64 units x 8 row lanes) adds its slice of a tile -- every load of a thread is independent,
// so the whole 8 MB of a 4096 x 64 x 7 evaluation is in flight at once -- and writes one partial row piece to scratch.
// k_pack_finish (the next launch on the stream: the kernel boundary makes the partial rows visible): one workgroup per tile adds the
// tile's partial rows in slice order (8 groups x slices / 8, then the groups in order); one more workgroup folds the per-wavefront cost
// sums (the association order of trk_reduce_sum) and traj_cost into out[0].  The association order never depends on the timing.
// Round 5: the one-launch form this replaces (a ticket per workgroup behind a __threadfence) took 46 us at 4096 x 64 x 7 and 146 us
// for config 5 -- 90 ns per WORKGROUP whatever it read (profiles/r05_exchange_trace_*_before.txt: 512 and 1920 workgroups): an
// agent-scope release on this chip writes back and invalidates the XCD's whole L2, and the workgroups' fences serialise.
// scratch: float[TRK_PACK_SLICES * C] (trk_pack_sums_scratch_bytes keeps a few spare words: earlier versions kept tickets there).
// ============================================================================================
#define TRK_PACK_SLICES 128
#define TRK_PACK_THREADS 512
// G: element type of gq (fp32, or the fp16 gradient of the reduced-precision rollout); unscale = 1 / grad_scale of that gradient, applied
// once to the fp32 column sums; traj_cost (nullable) [B]: a per-trajectory cost (the GP prior's) whose sum joins out[0].
template <class G, int V> struct PackVec;
template <> struct PackVec<float, 4> { typedef float4 T; static __device__ __forceinline__ void add(float (&a)[4], const T& v) { a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w; } };
template <> struct PackVec<float, 1> { typedef float T; static __device__ __forceinline__ void add(float (&a)[1], const T& v) { a[0] += v; } };
typedef _Float16 trk_half4 __attribute__((ext_vector_type(4)));
template <> struct PackVec<_Float16, 4> { typedef trk_half4 T; static __device__ __forceinline__ void add(float (&a)[4], const T& v) { a[0] += (float)v[0]; a[1] += (float)v[1]; a[2] += (float)v[2]; a[3] += (float)v[3]; } };
template <> struct PackVec<_Float16, 1> { typedef _Float16 T; static __device__ __forceinline__ void add(float (&a)[1], const T& v) { a[0] += (float)v; } };

template <class G, int V>
__global__ void __launch_bounds__(TRK_PACK_THREADS)
k_pack_partial(const float* __restrict__ cost, const G* __restrict__ gq, int B, int H, int D, int slices, float* __restrict__ scratch) {
    __shared__ float part[8][64][V];
    const int C = H + H * D;                                 // columns: H of the cost matrix, then H D of the gradient matrix
    const int CU = C / V;                                    // units (V divides H, hence C)
    const int tiles = (CU + 63) / 64;
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;   // unit in the tile, row lane 0..7
    const int u = tile * 64 + c;
    const int rows_per = (B + slices - 1) / slices;
    const int b0 = slice * rows_per, b1 = min(B, b0 + rows_per);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.0f;
    if (u < CU) {
        const int col = u * V;
        if (col < H) {
            typedef typename PackVec<float, V>::T T;
            const T* src = reinterpret_cast<const T*>(cost + col);
            const int64_t stride = H / V;
#pragma unroll 4
            for (int b = b0 + rl; b < b1; b += 8) PackVec<float, V>::add(acc, src[b * stride]);
        } else {
            typedef typename PackVec<G, V>::T T;
            const T* src = reinterpret_cast<const T*>(gq + (col - H));
            const int64_t stride = (int64_t)H * D / V;
#pragma unroll 4
            for (int b = b0 + rl; b < b1; b += 8) PackVec<G, V>::add(acc, src[b * stride]);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) part[rl][c][k] = acc[k];
    __syncthreads();
    if (rl == 0 && u < CU) {
#pragma unroll
        for (int k = 0; k < V; ++k)
            scratch[(size_t)slice * C + u * V + k] = ((part[0][c][k] + part[1][c][k]) + (part[2][c][k] + part[3][c][k])) +
                                                     ((part[4][c][k] + part[5][c][k]) + (part[6][c][k] + part[7][c][k]));
    }
}

template <int V>
__global__ void __launch_bounds__(TRK_PACK_THREADS)
k_pack_finish(const float* __restrict__ scratch, float unscale, const float* __restrict__ block_sums, const float* __restrict__ traj_cost,
              int B, int H, int D, int64_t nb, int slices, float* __restrict__ out) {
    __shared__ float part[8][64][V];
    const int C = H + H * D, CU = C / V;
    const int tile = blockIdx.x;
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int u = tile * 64 + c;
    if (tile == (int)gridDim.x - 1) {
        // the extra workgroup: out[0] = the per-wavefront cost sums (the association order of trk_reduce_sum) + traj_cost, beside the tiles
        float* flat = &part[0][0][0];                           // >= 512 floats
        float a = 0.0f;
        if (threadIdx.x < 256) {
            for (int64_t i = threadIdx.x; i < nb; i += 256) a += block_sums[i];
            if (traj_cost) { float t = 0.0f; for (int i = threadIdx.x; i < B; i += 256) t += traj_cost[i]; a += t; }
            flat[threadIdx.x] = a;
        }
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) flat[threadIdx.x] += flat[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[0] = flat[0];
        return;
    }
    // the tile's partial rows: row lane g adds the slices g * per .. (g + 1) * per - 1 in order, then the 8 groups in order
    const int per = (slices + 7) / 8;
    const int p0 = rl * per, p1 = min(slices, p0 + per);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.0f;
    if (u < CU) {
        typedef typename PackVec<float, V>::T T;
        const T* src = reinterpret_cast<const T*>(scratch + u * V);
        const int64_t stride = C / V;
#pragma unroll 8
        for (int p = p0; p < p1; ++p) PackVec<float, V>::add(acc, src[p * stride]);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) part[rl][c][k] = acc[k];
    __syncthreads();
    if (rl == 0 && u < CU) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float tot = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += part[g][c][k];
            const int col = u * V + k;
            out[1 + col] = col < H ? tot : tot * unscale;
        }
    }
}

// --------------------------------------------------------------------------------------------
// host-callable launchers (used by trk_capi.hip)
// --------------------------------------------------------------------------------------------
#include "trk_launch.h"

static inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

void trk_launch_fk_forward(int mode, const DevModelHdr& hdr, const DevLink* links, const SelMap& sel, int n_sel,
                           const float* q, int64_t n, float* out, hipStream_t st) {
    size_t lds = sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs + (size_t)hdr.n_slots * 12 * TRK_WAVE +
                                  (mode == 1 ? (size_t)TRK_WAVE * ((n_sel * 3) | 1) : (size_t)TRK_WAVE * 20));
    if (mode == 0) hipLaunchKernelGGL(k_fk_forward<0>, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links, sel, n_sel, q, n, out);
    else hipLaunchKernelGGL(k_fk_forward<1>, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links, sel, n_sel, q, n, out);
}

void trk_launch_fk_backward(int mode, const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const SelMap& sel,
                            const SelMap& selp, int n_sel, const float* q, const float* gin, int64_t n, float* gq, hipStream_t st) {
    size_t lds = sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * 8 + (size_t)hdr.n_slots * 12 * TRK_WAVE +
                                  (mode == 1 ? (size_t)TRK_WAVE * ((n_sel * 3) | 1) : 0));
    if (mode == 0) hipLaunchKernelGGL(k_fk_backward<0>, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links, fin, sel, selp, n_sel, q, gin, n, gq);
    else hipLaunchKernelGGL(k_fk_backward<1>, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links, fin, sel, selp, n_sel, q, gin, n, gq);
}

void trk_launch_ik_step(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, int link, const float* H_target,
                        int per_sample, const float* lower, const float* upper, float w_jl, float se3_eps, float lr,
                        const IkSchedule& sched, int n_steps, int64_t n, float* q, float* mom, float* vel, float* loss,
                        uint8_t* valid, hipStream_t st) {
    size_t lds = sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * 8 + (size_t)hdr.n_slots * 12 * TRK_WAVE);
    hipLaunchKernelGGL(k_ik_step, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links, fin, link, H_target,
                       per_sample, lower, upper, w_jl, se3_eps, lr, sched, n_steps, n, q, mom, vel, loss, valid);
}

void trk_launch_cost_fields(const DevCostHdr& C, int fields, const float* link_pos, int64_t n, const float* gcost,
                            float* cost, float* g_link_pos, hipStream_t st) {
    size_t lds = sizeof(float) * (g_link_pos ? 2 : 1) * (size_t)TRK_WAVE * (((C.n_links_in + C.n_virtual) * 3) | 1);
    hipLaunchKernelGGL(k_cost_fields, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, C, fields, link_pos, n, gcost, cost, g_link_pos);
}

void trk_launch_collision_fields(const DevCostHdr& C, int fields, const float* link_pos, int64_t n, float margin,
                                 int use_default, uint8_t* out, hipStream_t st) {
    size_t lds = sizeof(float) * (size_t)TRK_WAVE * (((C.n_links_in + C.n_virtual) * 3) | 1);
    hipLaunchKernelGGL(k_collision_fields, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, C, fields, link_pos, n, margin, use_default, out);
}

void trk_launch_ee_cost(const DevCostHdr& C, const float* H, int64_t n, int64_t stride, const float* target, int per_sample,
                        const float* gcost, float* cost, float* gH, int64_t g_stride, hipStream_t st) {
    hipLaunchKernelGGL(k_ee_cost, dim3(grid_for(n, 256)), dim3(256), 0, st, C, H, n, stride, target, per_sample, gcost, cost, gH, g_stride);
}

size_t trk_lds_rollout(const DevModelHdr& hdr, int n_cols) {
    return sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * 8 + (size_t)hdr.n_slots * 12 * TRK_WAVE +
                            2 * (size_t)TRK_WAVE * ((n_cols * 3) | 1));
}
size_t trk_lds_fk_points(const DevModelHdr& hdr, int n_points, bool backward) {
    return sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * (backward ? 8 : 1) + (size_t)hdr.n_slots * 12 * TRK_WAVE +
                            (size_t)TRK_WAVE * ((n_points * 3) | 1));
}

template <class IO, class G>
static void launch_rollout_generic(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const DevPointSet* ps,
                                   const DevCostHdr& C, const TrkRolloutWeights& w, const void* q, int64_t n, void* link_pos,
                                   float* cost, void* gq, float* cost_sum, float grad_scale, hipStream_t st) {
    const IO* qq = static_cast<const IO*>(q);
    IO* lp = static_cast<IO*>(link_pos);
    G* gg = static_cast<G*>(gq);
    if (ps) hipLaunchKernelGGL((k_rollout_generic<true, IO, G>), dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), trk_lds_rollout(hdr, ps->n_points + C.n_virtual), st,
                               hdr, links, fin, SelMap{}, *ps, C, w, qq, n, lp, cost, gg, cost_sum, grad_scale);
    else hipLaunchKernelGGL((k_rollout_generic<false, IO, G>), dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), trk_lds_rollout(hdr, hdr.n_links + C.n_virtual), st,
                            hdr, links, fin, SelMap{}, DevPointSet{}, C, w, qq, n, lp, cost, gg, cost_sum, grad_scale);
}

// io_mode: 0 fp32 | 1 fp16 q / link_pos / gq | 2 fp16 q / link_pos, fp32 gq (TRK_IO_* of trk_spec_common.h)
void trk_launch_rollout_generic(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const DevPointSet* ps,
                                const DevCostHdr& C, const TrkRolloutWeights& w, int io_mode, float grad_scale, const void* q, int64_t n,
                                void* link_pos, float* cost, void* gq, float* cost_sum, hipStream_t st) {
    if (io_mode == 1) launch_rollout_generic<_Float16, _Float16>(hdr, links, fin, ps, C, w, q, n, link_pos, cost, gq, cost_sum, grad_scale, st);
    else if (io_mode == 2) launch_rollout_generic<_Float16, float>(hdr, links, fin, ps, C, w, q, n, link_pos, cost, gq, cost_sum, grad_scale, st);
    else launch_rollout_generic<float, float>(hdr, links, fin, ps, C, w, q, n, link_pos, cost, gq, cost_sum, 1.0f, st);
}

void trk_launch_fk_points(const DevModelHdr& hdr, const DevLink* links, const DevPointSet& ps, const float* q, int64_t n,
                          float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_fk_points, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), trk_lds_fk_points(hdr, ps.n_points, false), st,
                       hdr, links, ps, q, n, out);
}

void trk_launch_fk_points_backward(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const DevPointSet& ps,
                                   const float* q, const float* gin, int64_t n, float* gq, hipStream_t st) {
    hipLaunchKernelGGL(k_fk_points_backward, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), trk_lds_fk_points(hdr, ps.n_points, true), st,
                       hdr, links, fin, ps, q, gin, n, gq);
}

void trk_launch_fk_jacobian(const DevModelHdr& hdr, const DevLink* links_dev, const DevLink* links_host, const float* q,
                            const float* qd, int64_t n, int link, int link_joint_idx, float* pos, float* quat,
                            float* lin_jac, float* ang_jac, float* vel_lin, float* vel_ang, hipStream_t st) {
    // which DOFs get a column (robot_tree.py:239-244, the reference's serial-chain rule applied to every visited link)
    JacCols cols;
    for (int d = 0; d < TRK_MAX_DOFS; ++d) cols.slot[d] = -1;
    cols.n_cols = 0; cols.p_end = 1;
    for (int p = 0; p < hdr.n_links; ++p) {
        const DevLink& Lk = links_host[p];
        if (Lk.link == link) cols.p_end = std::max(cols.p_end, p + 1);
        if (Lk.dof >= 0 && (Lk.link - 1) <= link_joint_idx && Lk.jac_axis >= 0) {
            cols.slot[Lk.dof] = (int8_t)cols.n_cols++;
            cols.p_end = std::max(cols.p_end, p + 1);
        }
    }
    const int rstride = (6 * cols.n_cols + 3) | 1;
    size_t lds = sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * (qd ? 2 : 1) + (size_t)hdr.n_slots * (qd ? 18 : 12) * TRK_WAVE +
                                  (size_t)TRK_WAVE * rstride + TRK_MAX_DOFS);
    hipLaunchKernelGGL(k_fk_jacobian, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links_dev, cols, q, qd, n, link,
                       link_joint_idx, pos, quat, lin_jac, ang_jac, vel_lin, vel_ang);
}

void trk_launch_fk_analytic_jacobian(const DevModelHdr& hdr, const DevLink* links, const void* dofs, const float* q,
                                     int64_t n, float* J, hipStream_t st) {
    size_t lds = sizeof(float) * ((size_t)TRK_WAVE * hdr.n_dofs * 7 + (size_t)hdr.n_slots * 12 * TRK_WAVE +
                                  (size_t)TRK_WAVE * ((7 * hdr.n_dofs) | 1));
    hipLaunchKernelGGL(k_fk_analytic_jacobian, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), lds, st, hdr, links,
                       static_cast<const DofRec*>(dofs), q, n, J);
}

void trk_launch_rotmat_to_quat(const float* R, int64_t n, int stride, int pitch, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_rotmat_to_quat, dim3(grid_for(n, 256)), dim3(256), 0, st, R, n, stride, pitch, out);
}

void trk_launch_frame_compose(int op, const float* Ra, const float* ta, int a_bcast, const float* Rb, const float* tb, int b_bcast,
                              int64_t n, float* Ro, float* to, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_compose, dim3(grid_for(n, 256)), dim3(256), 0, st, op, Ra, ta, a_bcast, Rb, tb, b_bcast, n, Ro, to);
}
void trk_launch_frame_compose_bwd(int op, const float* Ra, const float* ta, const float* Rb, const float* tb, const float* gR,
                                  const float* gt, int64_t n, float* gRa, float* gta, float* gRb, float* gtb, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_compose_bwd, dim3(grid_for(n, 256)), dim3(256), 0, st, op, Ra, ta, Rb, tb, gR, gt, n, gRa, gta,
                       gRb, gtb);
}
void trk_launch_frame_transform_points(const float* R, const float* t, int64_t n, const float* pts, int P, float* out,
                                       hipStream_t st) {
    hipLaunchKernelGGL(k_frame_transform_points, dim3(grid_for(n * P, 256)), dim3(256), 0, st, R, t, n, pts, P, out);
}
void trk_launch_frame_transform_points_bwd(const float* g, int64_t n, const float* pts, int P, float* gR, float* gt,
                                           hipStream_t st) {
    hipLaunchKernelGGL(k_frame_transform_points_bwd, dim3(grid_for(n, 256)), dim3(256), 0, st, g, n, pts, P, gR, gt);
}
void trk_launch_frame_quat_euler(const float* R, int64_t n, int stride, int pitch, float* quat_xyzw, float* euler,
                                 hipStream_t st) {
    hipLaunchKernelGGL(k_frame_quat_euler, dim3(grid_for(n, 256)), dim3(256), 0, st, R, n, stride, pitch, quat_xyzw, euler);
}
void trk_launch_frame_quat_euler_bwd(const float* R, int64_t n, int stride, int pitch, const float* gquat_xyzw, const float* geuler,
                                     float* gR, hipStream_t st) {
    hipLaunchKernelGGL(k_frame_quat_euler_bwd, dim3(grid_for(n, 256)), dim3(256), 0, st, R, n, stride, pitch, gquat_xyzw, geuler, gR);
}

void trk_launch_rotation_from(int axis, const float* in, int64_t n, float* R, const float* gR, float* gin, hipStream_t st) {
    hipLaunchKernelGGL(k_rotation_from, dim3(grid_for(n, 256)), dim3(256), 0, st, axis, in, n, R, gR, gin);
}

void trk_launch_grid_precompute(const DevCostHdr& C, const int32_t* dims, const float* lo, const float* hi, float* sdf,
                                float* grad, hipStream_t st) {
    const int64_t total = (int64_t)dims[0] * dims[1] * dims[2];
    hipLaunchKernelGGL(k_grid_precompute, dim3(grid_for(total, 256)), dim3(256), 0, st, C, dims[0], dims[1], dims[2],
                       lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], sdf, grad);
}

void trk_launch_interpolate(const float* x, int64_t T, int H, int D, int n_interp, const float* alpha, const float* beta,
                            float* out, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)H * D + 2 * (size_t)n_interp);
    if (lds <= 48 * 1024 && T <= 0x7fffffff) {
        hipLaunchKernelGGL(k_interpolate_via_points, dim3((unsigned)T), dim3(256), lds, st, x, T, H, D, n_interp, alpha, beta, out);
        return;
    }
    const int64_t total = T * (int64_t)(H - 1) * n_interp * D;
    hipLaunchKernelGGL(k_interpolate_via_points_flat, dim3(grid_for(total, 256)), dim3(256), 0, st, x, T, H, D, n_interp, alpha, beta, out);
}

void trk_launch_traj_validate(const uint8_t* wp, const float* x, int64_t T, int H, int S, int Hi, int D, const float* qmin,
                              const float* qmax, int64_t inner, uint8_t* flags, int64_t* idx, int32_t* counts,
                              int32_t* counts_host, int32_t ticket, float* gathered, hipStream_t st) {
    // Hi < 0: wp holds the per-wavefront partial flags of the via-point launch for trajectories of -Hi interpolated configurations
    const bool have_partial = Hi < 0;
    const int64_t hi = have_partial ? -(int64_t)Hi : 1;
    if (T > 0 && !have_partial) hipLaunchKernelGGL(k_traj_flags, dim3(grid_for(T, 4)), dim3(256), 0, st, wp, Hi, x, H, S, D, qmin, qmax, T, flags);
    hipLaunchKernelGGL(k_traj_partition, dim3(1), dim3(1024), 0, st, flags, have_partial ? wp : nullptr, hi, trk_via_slots(hi), T, inner, idx, counts,
                       counts_host, ticket);   // T == 0: zeros + ticket
    if (gathered && T > 0) hipLaunchKernelGGL(k_traj_gather, dim3((unsigned)T), dim3(128), 0, st, x, H * S, inner > 0 ? 2 : 1, idx, inner, gathered);
}

int trk_launch_jtj(int mfma, const float* lin, const float* ang, const float* r6, int64_t n, int D, float* JtJ, float* Jtr,
                   const float* damping, int damping_stride, float* dq, hipStream_t st) {
    if (mfma && D > 8) return -1;
    const size_t js = (size_t)(6 * (mfma ? 8 : D)) | 1, os = (size_t)(D * D + D) | 1;
    const bool fixed = !mfma && (D == 6 || D == 7) && std::getenv("TRK_EXP_JTJ_GENERIC") == nullptr;      // (the knob: same-box A/B against the runtime-D kernel)
    const size_t lds = sizeof(float) * TRK_JTJ_WAVES * TRK_WAVE * (fixed ? (js > os ? js : os) : js + os);
    if (lds > 160 * 1024) return -1;
    const dim3 grid(grid_for(n, TRK_JTJ_WAVES * TRK_WAVE)), block(TRK_JTJ_WAVES * TRK_WAVE);
    if (mfma) hipLaunchKernelGGL(k_jtj<true>, grid, block, lds, st, lin, ang, r6, n, D, JtJ, Jtr, damping, damping_stride, dq);
    else if (fixed && D == 7) hipLaunchKernelGGL((k_jtj<false, 7>), grid, block, lds, st, lin, ang, r6, n, D, JtJ, Jtr, damping, damping_stride, dq);
    else if (fixed) hipLaunchKernelGGL((k_jtj<false, 6>), grid, block, lds, st, lin, ang, r6, n, D, JtJ, Jtr, damping, damping_stride, dq);
    else hipLaunchKernelGGL(k_jtj<false>, grid, block, lds, st, lin, ang, r6, n, D, JtJ, Jtr, damping, damping_stride, dq);
    return 0;
}

void trk_launch_grid_pack(const float* sdf, const float* grad, const int32_t dims[3], int nb1, int nb2, int64_t n_rec, float4* cells, hipStream_t st) {
    hipLaunchKernelGGL(k_grid_pack, dim3(grid_for(n_rec, 256)), dim3(256), 0, st, sdf, grad, dims[0], dims[1], dims[2], nb1, nb2, n_rec, cells);
}

void trk_launch_scale_rows(int f16, const void* g, const float* sc, int sc_stride, int64_t n, int D, void* out, hipStream_t st) {
    if (f16) hipLaunchKernelGGL(k_scale_rows<_Float16>, dim3(grid_for(n * D, 256)), dim3(256), 0, st, (const _Float16*)g, sc, sc_stride, n, D, (_Float16*)out);
    else hipLaunchKernelGGL(k_scale_rows<float>, dim3(grid_for(n * D, 256)), dim3(256), 0, st, (const float*)g, sc, sc_stride, n, D, (float*)out);
}

void trk_launch_interpolate_columns(const float* x, int64_t n, int L, int C, int K, const int32_t* src, const float* w, float* out,
                                    hipStream_t st) {
    hipLaunchKernelGGL(k_interpolate_columns, dim3(grid_for(n * K * C, 256)), dim3(256), 0, st, x, n, L, C, K, src, w, out);
}
void trk_launch_interpolate_columns_bwd(const float* g, int64_t n, int L, int C, int K, const int32_t* src, const float* w, float* gx,
                                        hipStream_t st) {
    hipLaunchKernelGGL(k_interpolate_columns_bwd, dim3(grid_for(n * L * C, 256)), dim3(256), 0, st, g, n, L, C, K, src, w, gx);
}

int trk_launch_gp_prior(int f16, int grad_f16, float grad_scale, const void* q, const void* qd, int64_t B, int H, int D, float dt, float sigma, float w,
                        float* cost, void* gq, void* gqd, int accumulate, hipStream_t st) {
    const float s2 = 1.0f / (sigma * sigma);
    const float a = 12.0f * s2 / (dt * dt * dt), b = -6.0f * s2 / (dt * dt), c = 4.0f * s2 / dt;
    const size_t total = (size_t)H * D;
    const size_t lds = sizeof(float) * (2 * ((total + 3) & ~(size_t)3) + 4);
    if (lds > 160 * 1024) return -1;
    const size_t esz = f16 ? 2 : 4, gsz = grad_f16 ? 2 : 4;
    const uintptr_t in_ptrs = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(qd);
    const uintptr_t out_ptrs = reinterpret_cast<uintptr_t>(gq) | reinterpret_cast<uintptr_t>(gqd);
    const bool vec = total % 4 == 0 && (in_ptrs & (4 * esz - 1)) == 0 && (out_ptrs & (4 * gsz - 1)) == 0;
#define TRK_GP(T, G, V) hipLaunchKernelGGL((k_gp_prior<T, G, V>), dim3((unsigned)B), dim3(256), lds, st, (const T*)q, (const T*)qd, H, D, \
                                           dt, a, b, c, w, grad_scale, cost, (G*)gq, (G*)gqd, accumulate)
    if (f16 && grad_f16) { if (vec) TRK_GP(_Float16, _Float16, true); else TRK_GP(_Float16, _Float16, false); }
    else if (f16) { if (vec) TRK_GP(_Float16, float, true); else TRK_GP(_Float16, float, false); }
    else if (grad_f16) return -2;       // fp32 trajectories with an fp16 gradient: not a mode
    else { if (vec) TRK_GP(float, float, true); else TRK_GP(float, float, false); }
#undef TRK_GP
    return 0;
}

void trk_launch_gp_sample_cost(int f16, const void* q, const void* qd, int64_t n, int H, int D, float dt, float sigma, float w, float* cost,
                               float* block_sums, hipStream_t st) {
    const float s2 = 1.0f / (sigma * sigma);
    const float a = 12.0f * s2 / (dt * dt * dt), b = -6.0f * s2 / (dt * dt), c = 4.0f * s2 / dt;
    if (f16) hipLaunchKernelGGL(k_gp_sample_cost<_Float16>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const _Float16*)q, (const _Float16*)qd, n, H, D, dt, a, b, c, w, cost);
    else hipLaunchKernelGGL(k_gp_sample_cost<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const float*)q, (const float*)qd, n, H, D, dt, a, b, c, w, cost);
    if (block_sums) hipLaunchKernelGGL(k_block_sums, dim3(grid_for(n, TRK_WAVE)), dim3(TRK_WAVE), 0, st, cost, n, block_sums);
}

void trk_launch_finite_difference(const float* x, int64_t B, int H, int D, float dt, int method, float* out, hipStream_t st) {
    const int64_t total = B * H * D;
    hipLaunchKernelGGL(k_finite_difference, dim3(grid_for(total, 256)), dim3(256), 0, st, x, total, H, D, dt, method, out);
}

void trk_launch_traj_diff_norm_sum(const float* x, int64_t B, int H, int S, int c0, int D, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_traj_diff_norm_sum, dim3((unsigned)B), dim3(256), 0, st, x, H, S, c0, D, out);
}

size_t trk_pack_scratch_floats(int H, int D) { const size_t C = H + (size_t)H * D; return (size_t)TRK_PACK_SLICES * C + (C + 63) / 64 + 1; }
template <class G, int V>
static void launch_pack(const float* cost, const void* gq, float unscale, const float* block_sums, const float* traj_cost, int B, int H, int D,
                        int64_t nb, float* scratch, float* out, hipStream_t st) {
    const int C = H + H * D, tiles = (C / V + 63) / 64;
    const int slices = max(1, min(TRK_PACK_SLICES, (B + 7) / 8));
    hipLaunchKernelGGL((k_pack_partial<G, V>), dim3(slices * tiles), dim3(TRK_PACK_THREADS), 0, st, cost, static_cast<const G*>(gq), B, H, D, slices, scratch);
    hipLaunchKernelGGL((k_pack_finish<V>), dim3(tiles + 1), dim3(TRK_PACK_THREADS), 0, st, scratch, unscale, block_sums, traj_cost, B, H, D, nb, slices, out);
}
void trk_launch_pack_sums(const float* cost, const void* gq, int grad_f16, float unscale, const float* block_sums, const float* traj_cost,
                          int B, int H, int D, int64_t nb, float* scratch, float* out, hipStream_t st) {
    // the 16-byte (fp16: 8-byte) unit path needs whole units per row and aligned rows
    const bool vec = (H % 4 == 0) && (reinterpret_cast<uintptr_t>(cost) % 16 == 0) && (reinterpret_cast<uintptr_t>(scratch) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(gq) % (grad_f16 ? 8 : 16) == 0);
    if (grad_f16) { if (vec) launch_pack<_Float16, 4>(cost, gq, unscale, block_sums, traj_cost, B, H, D, nb, scratch, out, st);
                    else launch_pack<_Float16, 1>(cost, gq, unscale, block_sums, traj_cost, B, H, D, nb, scratch, out, st); }
    else { if (vec) launch_pack<float, 4>(cost, gq, unscale, block_sums, traj_cost, B, H, D, nb, scratch, out, st);
           else launch_pack<float, 1>(cost, gq, unscale, block_sums, traj_cost, B, H, D, nb, scratch, out, st); }
}

void trk_launch_reduce_sum(const float* x, int64_t n, float* out, hipStream_t st) {
    if (n > TRK_REDUCE_WIDE_FROM) hipLaunchKernelGGL(k_reduce_sum_wide, dim3(1), dim3(1024), 0, st, x, n, out);
    else hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(256), 0, st, x, n, out);
}

void trk_launch_sdf_points(const DevCostHdr& C, const float* pts, int64_t n, float* sdf, float* grad, hipStream_t st) {
    hipLaunchKernelGGL(k_sdf_points, dim3(grid_for(n, 256)), dim3(256), 0, st, C, pts, n, sdf, grad);
}

int trk_kernels_init(void) {
    const int max_lds = 160 * 1024;
    hipError_t e = hipSuccess;
#define TRK_SET(k) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)
    TRK_SET(k_fk_forward<0>); TRK_SET(k_fk_forward<1>); TRK_SET(k_fk_backward<0>); TRK_SET(k_fk_backward<1>);
    TRK_SET(k_cost_fields); TRK_SET(k_collision_fields); TRK_SET((k_rollout_generic<false, float>)); TRK_SET((k_rollout_generic<true, float>));
    TRK_SET((k_rollout_generic<false, _Float16>)); TRK_SET((k_rollout_generic<true, _Float16>));
    TRK_SET((k_rollout_generic<false, _Float16, float>)); TRK_SET((k_rollout_generic<true, _Float16, float>));
    TRK_SET(k_fk_jacobian); TRK_SET(k_fk_points); TRK_SET(k_fk_points_backward);
    TRK_SET(k_fk_analytic_jacobian); TRK_SET(k_ik_step);
    TRK_SET((k_gp_prior<float, float, true>)); TRK_SET((k_gp_prior<float, float, false>));
    TRK_SET((k_gp_prior<_Float16, _Float16, true>)); TRK_SET((k_gp_prior<_Float16, _Float16, false>));
    TRK_SET((k_gp_prior<_Float16, float, true>)); TRK_SET((k_gp_prior<_Float16, float, false>));
    TRK_SET(k_jtj<true>); TRK_SET(k_jtj<false>);
#undef TRK_SET
    return e == hipSuccess ? 0 : -1;
}
