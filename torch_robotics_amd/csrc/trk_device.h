// trk_device.h -- device-side data layout and math shared by the gfx950 kernels.
//
// Model and cost tables are wave-uniform: kernels index them with loop counters, so the
// compiler fetches them with scalar loads (s_load_dwordx*) into SGPRs and feeds them to VALU
// ops as scalar operands -- no VGPRs, no LDS traffic.  LDS is reserved for what is per-lane:
// the q / gq / link-position transposes that make HBM accesses coalesced, the pose stack of
// branched trees, and the per-joint records of the reverse pass.
#pragma once
#ifndef __HIPCC_RTC__                  // hipRTC (jit.py's fall-back when hipcc is absent) brings its own runtime header
#include <hip/hip_runtime.h>
#include <stdint.h>
#else                                  // ... but no <stdint.h> names in the global namespace
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#endif
#include "../../include/trk.h"

#define TRK_WAVE 64

// One record per link, stored by DFS pre-order position (32 dwords = 128 B, one s_load_dwordx16 pair).
struct alignas(16) DevLink {
    float Rf[9];        // R_fixed row-major
    float trans[3];
    float axis[3];      // raw <axis>
    float lower, upper;
    float rot_sign;
    int32_t type;       // TrkJointType
    int32_t dof;        // DOF index or -1
    int32_t rot_axis;   // stateless rotation axis
    int32_t clamp;
    int32_t parent_slot;
    int32_t store_slot;
    int32_t link;       // file-order link index
    int32_t sf_rot_axis;
    int32_t sf_clamp;
    int32_t jac_axis;
    int32_t fin_begin;  // joints whose subtree ends after this position: fin[fin_begin, fin_end)
    int32_t fin_end;
    int32_t _pad[2];
};
static_assert(sizeof(DevLink) == 128, "DevLink must be 32 dwords");

struct DevModelHdr {
    int32_t n_links, n_dofs, n_slots, _pad;
    float base_R[9];
    float base_t[3];
};

struct alignas(16) DevPrim {   // 8 dwords
    int32_t type;
    float cx, cy, cz, hx, hy, hz, r;
};

struct alignas(16) DevObj {    // 16 dwords
    float pos[3];
    float R[9];
    int32_t prim_begin, prim_end, is_grid, identity;   // identity: bit0 = R is I, bit1 = has non-sphere primitives
};
#define TRK_OBJ_IDENTITY 1
#define TRK_OBJ_NONSPHERE 2
// per-object flag stored in DevPrim.type's upper bits is avoided: sphere primitives are skipped by type in
// the cost kernels (they live in DevCostHdr::spheres), and visited only by the per-object SDF queries.

struct DevGrid {
    // bricks along y and z (the record table is tiled, see grid_cell); two spare words keep the struct's size
    int32_t nb1, nb2;
    int32_t _pad[2];
    const float4* cells;        // (gx, gy, gz, sdf) per voxel, built at trk_cost_model_create: ONE 16-byte gather per point instead
                                // of a 4- and a 12-byte one from two arrays (random cells: every gather is its own cache line)
    int32_t dims[3];
    float lim_min[3];
    float map_dim[3];
    float fdims[3];
};

// Passed to kernels by value (kernarg segment, scalar-loaded).
struct DevCostHdr {
    int32_t n_links_in;
    int32_t n_obj_links, n_objects, has_grid, has_ws;
    int32_t n_self_links, n_self_pairs;
    int32_t ee_link, ee_square;
    float ee_w_pos, ee_w_rot;
    float ws_min[3], ws_max[3];
    float ws_c[3], ws_h[3];        // box centre and half widths (host, fp32): distance to the nearest plane of axis k = h_k - |p_k - c_k|
    float ee_target[16];
    int32_t ee2_link; int32_t _pad_ee2;   // second tracked link (-1 = none), same weights
    float ee2_target[16];
    const int32_t* obj_link_idx;   // device
    const float* obj_link_margin;  // device
    const DevObj* objects;         // device
    const DevPrim* prims;          // device
    const int32_t* self_pairs;     // device: pairs already mapped to position-tensor link indices [P*2]
    const float* self_margin;      // device
    // All sphere primitives of all analytic objects, merged and moved to the world frame (a sphere SDF is
    // rotation-invariant, so the object pose folds into the centre; max over objects of (margin - sdf) is a
    // min over the union).  float4 = (cx, cy, cz, r).  Objects keep only their non-sphere primitives here.
    const float4* spheres;         // device
    const float4* spheres_sel;     // device: (-2cx, -2cy, -2cz, |c|^2): |p-c|^2 - |p|^2 = p . sel.xyz + sel.w (3 FMAs)
    int32_t n_spheres;
    int32_t spheres_uniform_r;     // 1: every radius equals sphere_r (arg-min by squared distance, one sqrt)
    float sphere_r;
    int32_t n_box_objects;         // objects that still have non-sphere primitives (not the grid)
    const int32_t* box_objects;    // device: their indices into objects[]
    DevGrid grid;
    // spheres_sel again, two spheres per record: [Sx Tx | Sy Ty | Sz Tz | Sw Tw] (S = sphere 2j, T = sphere 2j + 1).  An odd
    // table is padded with a copy of its last sphere (spheres[] / spheres_sel[] hold the copy too, n_spheres does not count it).
    const float* sphere_pairs;     // device
    int32_t n_sphere_pairs;
    int32_t clamp_fields;          // TRK_FIELD_* mask: relu(margin - sdf) per link / pair (clamp_sdf=True, distance_fields.py:114-117)
    // interpolate_link_pos (distance_fields.py:66-69, 145-147): position columns n_links_in + k = w[2k] * column src[2k] +
    // w[2k+1] * column src[2k+1]; the index tables above may name them.  Only the table-driven kernels evaluate them.
    int32_t n_virtual;
    int32_t self_single;           // 1: some self pair is (a, a) -- the single-link distance 1e9 |p|_1 (distance_fields.py:195-198)
    const int32_t* virtual_src;    // device [2 * n_virtual]
    const float* virtual_w;        // device [2 * n_virtual]
    int32_t n_prims;               // entries of prims[] (a fused kernel keeps up to TRK_LDS_PRIMS of them in LDS)
    int32_t _pad_prims;
};

// Points rigidly attached to links (grasped-object points robot_panda.py:154-168, per-link collision spheres):
// world position = R_link * off + t_link (Frame.transform_point frame.py:116-118).  pts[] is sorted by the pre-order
// position of the owning link; begin[p] .. begin[p+1] is the range of position p; col = output column.
struct alignas(16) DevPoint { float off[3]; int32_t col; };
struct DevPointSet {
    const DevPoint* pts;        // device
    const int32_t* begin;       // device [n_links + 1]
    int32_t n_points;
    int32_t _pad;
};

struct SelMap {                 // link (file index) -> output column, -1 = not selected
    int32_t col[TRK_MAX_LINKS];
};

// Wave-uniform tables are read through the constant address space: the loads are then invariant, so the
// compiler emits scalar loads (s_load_dwordx*, scalar cache -> SGPRs) instead of a vector load + full
// s_waitcnt vmcnt(0) per loop iteration.  The tables are never written by a kernel.
#define TRK_CAS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const TRK_CAS T* cptr(const T* p) { return (const TRK_CAS T*)p; }

__device__ __forceinline__ DevPrim load_prim(const DevPrim* prims, int i) {
    const TRK_CAS DevPrim* c = cptr(prims) + i;
    DevPrim P;
    P.type = c->type; P.cx = c->cx; P.cy = c->cy; P.cz = c->cz; P.hx = c->hx; P.hy = c->hy; P.hz = c->hz; P.r = c->r;
    return P;
}
__device__ __forceinline__ DevObj load_obj(const DevObj* objs, int i) {
    const TRK_CAS DevObj* c = cptr(objs) + i;
    DevObj O;
#pragma unroll
    for (int k = 0; k < 3; ++k) O.pos[k] = c->pos[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) O.R[k] = c->R[k];
    O.prim_begin = c->prim_begin; O.prim_end = c->prim_end; O.is_grid = c->is_grid; O.identity = c->identity;
    return O;
}
struct F8 { float v[8]; };
__device__ __forceinline__ F8 load_f8_uniform(const __attribute__((address_space(4))) float* p, int i) {   // one s_load_dwordx8
    F8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = p[8 * i + k];
    return r;
}
struct F4 { float x, y, z, w; };
__device__ __forceinline__ F4 load_f4_uniform(const float4* p, int i) {      // wave-uniform index: scalar load
    const TRK_CAS float* c = (const TRK_CAS float*)p + 4 * i;
    F4 v; v.x = c[0]; v.y = c[1]; v.z = c[2]; v.w = c[3];
    return v;
}

// ------------------------------------------------------------------------------------------
// math
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float trk_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, 1 ulp
__device__ __forceinline__ float trk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32, 1 ulp
__device__ __forceinline__ float trk_rsq(float x) { return __builtin_amdgcn_rsqf(x); }     // v_rsq_f32, 1 ulp

// Neighbour exchange along the wavefront as a DPP operand (gfx9 `wave_shl:1` / `wave_shr:1`; tools/microbench/dpp_wave_shift.hip): lane l
// receives v of lane l + 1 (l - 1); the lane without a source -- 63 (0) -- keeps `edge`.  All 64 lanes must be active.
__device__ __forceinline__ float trk_dpp_from_next(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float trk_dpp_from_prev(float edge, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}

// Sum over the wavefront, broadcast to every lane.  DPP adds (row_shr 1/2/4/8, row_bcast 15/31) run at VALU speed; the
// __shfl_xor butterfly is six dependent ds_bpermute round trips (~100 cycles each) at the very end of the kernel, where no
// other work is left to hide them.  Deterministic (fixed association), result taken from lane 63.
__device__ __forceinline__ float trk_wave_sum(float v) {
    int x = __float_as_int(v);
#define TRK_DPP_ADD(ctrl, rmask)                                                                          \
    x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true)))
    TRK_DPP_ADD(0x111, 0xf);      // row_shr:1
    TRK_DPP_ADD(0x112, 0xf);      // row_shr:2
    TRK_DPP_ADD(0x114, 0xf);      // row_shr:4
    TRK_DPP_ADD(0x118, 0xf);      // row_shr:8  -> lane 15 of each row holds the row's sum
    TRK_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    TRK_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef TRK_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// sin and cos of x together.  Cody-Waite reduction by pi (two fmas): x = k*pi + r, r in [-pi/2, pi/2], so
// sin x = (-1)^k sin r and cos x = (-1)^k cos r -- a sign flip (one shift, two xors), no sin/cos swap.
// sin r = r + r^3 P(r^2), cos r = 1 - r^2/2 + r^4 Q(r^2), P and Q cubic near-minimax fits (Chebyshev nodes).
// Max abs error 1.5e-7 (sin) / 1.3e-7 (cos) for |x| < 3000 rad against fp64 (tests/test_device_math_cpu.py);
// joint angles are clamped to URDF limits (a few rad).  20 VALU instructions, 13.5 per angle in the paired form.
#define TRK_INV_PI 0x1.45f306p-2f
#define TRK_PI_HI 0x1.921fb6p+1f
#define TRK_PI_LO -0x1.777a5cp-24f
#define TRK_S0 -0x1.555554p-3f
#define TRK_S1 0x1.11104ep-7f
#define TRK_S2 -0x1.9fb672p-13f
#define TRK_S3 0x1.619f8p-19f
#define TRK_C0 0x1.555556p-5f
#define TRK_C1 -0x1.6c163ep-10f
#define TRK_C2 0x1.9fd75ap-16f
#define TRK_C3 -0x1.1d07b4p-22f

__host__ __device__ __forceinline__ void trk_sincos(float x, float* s_out, float* c_out) {
    const float k = rintf(x * TRK_INV_PI);
    float r = fmaf(-k, TRK_PI_HI, x);
    r = fmaf(-k, TRK_PI_LO, r);
    const float u = r * r;
    float p = fmaf(u, TRK_S3, TRK_S2);
    p = fmaf(p, u, TRK_S1);
    p = fmaf(p, u, TRK_S0);
    const float s = fmaf(r * u, p, r);
    float q = fmaf(u, TRK_C3, TRK_C2);
    q = fmaf(q, u, TRK_C1);
    q = fmaf(q, u, TRK_C0);
    const float c = fmaf(u * u, q, fmaf(u, -0.5f, 1.0f));
    const unsigned flip = ((unsigned)(int)k) << 31;
    unsigned sb, cb;
    __builtin_memcpy(&sb, &s, 4); __builtin_memcpy(&cb, &c, 4);
    sb ^= flip; cb ^= flip;
    __builtin_memcpy(s_out, &sb, 4); __builtin_memcpy(c_out, &cb, 4);
}

#if defined(__HIPCC__)
typedef float trk_f2 __attribute__((ext_vector_type(2)));
typedef unsigned trk_u2 __attribute__((ext_vector_type(2)));
// two angles at once: the elementwise float2 arithmetic maps onto v_pk_mul_f32 / v_pk_fma_f32
__device__ __forceinline__ void trk_sincos2(float x0, float x1, float* s0, float* c0, float* s1, float* c1) {
    const trk_f2 x = {x0, x1};
    trk_f2 k = x * TRK_INV_PI;
    k.x = rintf(k.x); k.y = rintf(k.y);
    trk_f2 r = -k * TRK_PI_HI + x;
    r = -k * TRK_PI_LO + r;
    const trk_f2 u = r * r;
    trk_f2 p = u * TRK_S3 + TRK_S2;
    p = p * u + TRK_S1;
    p = p * u + TRK_S0;
    const trk_f2 s = (r * u) * p + r;
    trk_f2 q = u * TRK_C3 + TRK_C2;
    q = q * u + TRK_C1;
    q = q * u + TRK_C0;
    const trk_f2 c = (u * u) * q + (u * -0.5f + 1.0f);
    const unsigned f0 = ((unsigned)(int)k.x) << 31, f1 = ((unsigned)(int)k.y) << 31;
    *s0 = __uint_as_float(__float_as_uint(s.x) ^ f0); *c0 = __uint_as_float(__float_as_uint(c.x) ^ f0);
    *s1 = __uint_as_float(__float_as_uint(s.y) ^ f1); *c1 = __uint_as_float(__float_as_uint(c.y) ^ f1);
}
#endif

struct Pose {
    float r[9];
    float t[3];
};

__device__ __forceinline__ void pose_from_base(const DevModelHdr& h, Pose& p) {
#pragma unroll
    for (int k = 0; k < 9; ++k) p.r[k] = h.base_R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.t[k] = h.base_t[k];
}

// Rotate columns (i, j) of a 3x3 (row-major) by angle with sine s and cosine c:
// col_i' = c*col_i + s*col_j ; col_j' = -s*col_i + c*col_j   (= right-multiplication by Rot_k, k = the third axis)
__device__ __forceinline__ void rot_cols(float* r, int i, int j, float s, float c) {
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        const float a = r[3 * row + i], b = r[3 * row + j];
        r[3 * row + i] = fmaf(c, a, s * b);
        r[3 * row + j] = fmaf(c, b, -s * a);
    }
}

// child = parent o joint(link, q).  Stateless path: rigid_body.py:153-190 + geometrics/utils.py:11-17.
// `stateful` selects the quirks of update_joint_state (rigid_body.py:214-256).
// Returns the clamp mask (1.0f = gradient passes, 0.0f = q outside the limits).
template <bool STATEFUL>
__device__ __forceinline__ float joint_compose(const DevLink& L, const Pose& par, float q, Pose& out) {
    float pass = 1.0f;
    float qh = q;
    const bool is_joint = L.type != TRK_JOINT_FIXED;
    const int do_clamp = STATEFUL ? L.sf_clamp : L.clamp;
    if (is_joint && do_clamp) {
        pass = (q >= L.lower && q <= L.upper) ? 1.0f : 0.0f;
        qh = fminf(fmaxf(q, L.lower), L.upper);
    }
    // translation: t = R_p (trans [+ axis*q]) + t_p
    float tl0 = L.trans[0], tl1 = L.trans[1], tl2 = L.trans[2];
    if (L.type == TRK_JOINT_PRISMATIC) {
        tl0 = fmaf(L.axis[0], qh, tl0); tl1 = fmaf(L.axis[1], qh, tl1); tl2 = fmaf(L.axis[2], qh, tl2);
    }
#pragma unroll
    for (int row = 0; row < 3; ++row)
        out.t[row] = fmaf(par.r[3 * row], tl0, fmaf(par.r[3 * row + 1], tl1, fmaf(par.r[3 * row + 2], tl2, par.t[row])));
    // rotation: A = R_p R_fixed, then the joint rotation acts on two columns of A
    float A[9];
#pragma unroll
    for (int row = 0; row < 3; ++row)
#pragma unroll
        for (int col = 0; col < 3; ++col)
            A[3 * row + col] = fmaf(par.r[3 * row], L.Rf[col],
                                    fmaf(par.r[3 * row + 1], L.Rf[3 + col], par.r[3 * row + 2] * L.Rf[6 + col]));
    if (L.type == TRK_JOINT_REVOLUTE || L.type == TRK_JOINT_CONTINUOUS) {
        float s, c;
        trk_sincos(STATEFUL ? qh : L.rot_sign * qh, &s, &c);
        const int ax = STATEFUL ? L.sf_rot_axis : L.rot_axis;
        if (ax == 2) rot_cols(A, 0, 1, s, c);
        else if (ax == 0) rot_cols(A, 1, 2, s, c);
        else rot_cols(A, 2, 0, s, c);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) out.r[k] = A[k];
    return pass;
}

// ------------------------------------------------------------------------------------------
// signed distance fields (object frame), value + gradient
// ------------------------------------------------------------------------------------------
// sphere primitives.py:108-112, rounded box :327-334, sharp box :220-223
template <bool PRECISE>
__device__ __forceinline__ float prim_sdf(const DevPrim& P, float x, float y, float z, float& gx, float& gy, float& gz) {
    const float dx = x - P.cx, dy = y - P.cy, dz = z - P.cz;
    if (P.type == TRK_PRIM_SPHERE) {
        const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        const float nrm = PRECISE ? sqrtf(n2) : trk_sqrt(n2);
        const float inv = nrm > 0.0f ? (PRECISE ? 1.0f / nrm : trk_rcp(nrm)) : 0.0f;
        gx = dx * inv; gy = dy * inv; gz = dz * inv;
        return nrm - P.r;
    }
    const float sx = dx > 0.0f ? 1.0f : (dx < 0.0f ? -1.0f : 0.0f);
    const float sy = dy > 0.0f ? 1.0f : (dy < 0.0f ? -1.0f : 0.0f);
    const float sz = dz > 0.0f ? 1.0f : (dz < 0.0f ? -1.0f : 0.0f);
    const float ux = fabsf(dx) - P.hx + P.r, uy = fabsf(dy) - P.hy + P.r, uz = fabsf(dz) - P.hz + P.r;
    // arg-max with "first maximum wins" (torch.max / amax on CPU)
    int am = 0; float mu = ux;
    if (uy > mu) { mu = uy; am = 1; }
    if (uz > mu) { mu = uz; am = 2; }
    if (P.type == TRK_PRIM_SHARP_BOX) {          // P.r == 0 for sharp boxes
        gx = am == 0 ? sx : 0.0f; gy = am == 1 ? sy : 0.0f; gz = am == 2 ? sz : 0.0f;
        return mu;
    }
    const float rx = fmaxf(ux, 0.0f), ry = fmaxf(uy, 0.0f), rz = fmaxf(uz, 0.0f);
    const float n2 = fmaf(rx, rx, fmaf(ry, ry, rz * rz));
    const float nn = PRECISE ? sqrtf(n2) : trk_sqrt(n2);
    const float inv = nn > 0.0f ? (PRECISE ? 1.0f / nn : trk_rcp(nn)) : 0.0f;
    const float inside = mu < 0.0f ? 1.0f : 0.0f;
    gx = (rx * inv + (am == 0 ? inside : 0.0f)) * sx;
    gy = (ry * inv + (am == 1 ? inside : 0.0f)) * sy;
    gz = (rz * inv + (am == 2 ? inside : 0.0f)) * sz;
    return fminf(mu, 0.0f) + nn - P.r;
}

// position of voxel (i, j, k) in the tiled record table (host and device: trk_cost_model_create packs with the same function)
__host__ __device__ __forceinline__ int64_t grid_record(int nb1, int nb2, int i, int j, int k) {
    const int64_t brick = ((int64_t)(i >> 2) * nb1 + (j >> 2)) * nb2 + (k >> 2);
    const int in = ((i & 2) << 4) | ((j & 2) << 3) | ((k & 2) << 2) | ((i & 1) << 2) | ((j & 1) << 1) | (k & 1);
    return brick * 64 + in;
}
// grid_map_sdf.py:84-114: nearest-lower cell, stored gradient
typedef float trk_f3u __attribute__((ext_vector_type(3), aligned(4)));       // a 12-byte gradient record: one dwordx3 load
__device__ __forceinline__ int64_t grid_cell(const DevGrid& G, float x, float y, float z) {
    const float p[3] = {x, y, z};
    int idx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float f = floorf((p[k] - G.lim_min[k]) / G.map_dim[k] * G.fdims[k]);
        int v = (int)f;
        v = v < 0 ? 0 : v;
        v = v > G.dims[k] - 1 ? G.dims[k] - 1 : v;
        idx[k] = v;
    }
    // The records are TILED: 4 x 4 x 4 bricks of 1 KiB, inside a brick 2 x 2 x 2 cubes of 128 bytes (one L2 line).  A planner's
    // trajectories are smooth along the horizon -- the 64 lanes of a wavefront are consecutive time steps --, so neighbouring lanes
    // ask for neighbouring cells: in x-major order a step along x or y lands 640 KB / 3.2 KB away, here it stays in the line or
    // the brick.  (The values gathered are the same: nearest-lower cell, stored gradient.)
    return grid_record(G.nb1, G.nb2, idx[0], idx[1], idx[2]);
}
__device__ __forceinline__ float grid_sdf(const DevGrid& G, float x, float y, float z, float& gx, float& gy, float& gz) {
    const float4 c = G.cells[grid_cell(G, x, y, z)];
    gx = c.x; gy = c.y; gz = c.z;
    return c.w;
}

// ObjectField primitives.py:387-405: x' = R^T (x - pos), min over primitives, g = R g'.
// SKIP_SPHERES: visit only the non-sphere primitives (the spheres are handled through DevCostHdr::spheres);
// returns +inf when the object has none.
template <bool PRECISE, bool SKIP_SPHERES = false>
__device__ __forceinline__ float object_sdf(const DevCostHdr& C, int o, float x, float y, float z,
                                            float& gx, float& gy, float& gz) {
    const DevObj O = load_obj(C.objects, o);
    if (!SKIP_SPHERES && O.is_grid) return grid_sdf(C.grid, x, y, z, gx, gy, gz);
    const bool ident = (O.identity & TRK_OBJ_IDENTITY) != 0;
    float lx, ly, lz;
    if (ident) { lx = x - O.pos[0]; ly = y - O.pos[1]; lz = z - O.pos[2]; }
    else {
        const float dx = x - O.pos[0], dy = y - O.pos[1], dz = z - O.pos[2];
        lx = fmaf(O.R[0], dx, fmaf(O.R[3], dy, O.R[6] * dz));
        ly = fmaf(O.R[1], dx, fmaf(O.R[4], dy, O.R[7] * dz));
        lz = fmaf(O.R[2], dx, fmaf(O.R[5], dy, O.R[8] * dz));
    }
    float best = __builtin_inff(), bx = 0.0f, by = 0.0f, bz = 0.0f;
    for (int pi = O.prim_begin; pi < O.prim_end; ++pi) {
        const DevPrim P = load_prim(C.prims, pi);
        if (SKIP_SPHERES && P.type == TRK_PRIM_SPHERE) continue;
        float px, py, pz;
        const float v = prim_sdf<PRECISE>(P, lx, ly, lz, px, py, pz);
        const bool take = v < best;
        best = take ? v : best; bx = take ? px : bx; by = take ? py : by; bz = take ? pz : bz;
    }
    if (ident) { gx = bx; gy = by; gz = bz; }
    else {
        gx = fmaf(O.R[0], bx, fmaf(O.R[1], by, O.R[2] * bz));
        gy = fmaf(O.R[3], bx, fmaf(O.R[4], by, O.R[5] * bz));
        gz = fmaf(O.R[6], bx, fmaf(O.R[7], by, O.R[8] * bz));
    }
    return best;
}

// min over the whole scene (merged spheres, then objects with non-sphere primitives / the grid) of the signed
// distance at NL points held in registers, with the world-frame gradient of the arg-min primitive.
// = min_o sdf_o(p)  of distance_fields.py:307-316 + :121-122 (max over objects of margin - sdf).
// Tick slots the scene evaluation owns: one per sphere pair for the first pairs.  The generator numbers the position chunks
// at compile time, so this must equal codegen.OBJ_TICK_SLOTS.  Same-box A/B on the headline workload (4096 x 64, 9 chunks,
// 5 pairs): 8 slots 10.58 us (the three slots behind the last pair fire back to back), 6 -> 10.42, 5 -> 10.21, 4 -> 10.27,
// 3 -> 10.35; chunks issued right after staging cost +0.35 us.  The store pipe wants a smooth, late trickle: a wave that
// issues faster than HBM drains just sits at its next store.
#ifndef TRK_OBJ_TICK_SLOTS
#define TRK_OBJ_TICK_SLOTS 5
#endif
struct NoTick { template <int J> __device__ __forceinline__ void at() const {} };
#ifndef TRK_EXP_UNROLLED_RANK
#define TRK_EXP_UNROLLED_RANK 0     // experiment (tools/ab_defines_points.sh): the guarded, unrolled pair ranking for the NoTick callers too
#endif

template <class T> struct TickIsNoTick { static constexpr bool value = false; };
template <> struct TickIsNoTick<NoTick> { static constexpr bool value = true; };
template <> struct TickIsNoTick<const NoTick&> { static constexpr bool value = true; };
template <> struct TickIsNoTick<NoTick&> { static constexpr bool value = true; };
template <int J, class Tick>
__device__ __forceinline__ void scene_all_ticks(const Tick& tick) {
    if constexpr (J < TRK_OBJ_TICK_SLOTS) { tick.template at<J>(); scene_all_ticks<J + 1>(tick); }
}
// One guarded trip of the pair-ranking loop (see scene_min_sdf): pair J of at most 8.
template <int NL, int J, class Tick>
__device__ __forceinline__ void scene_rank_pairs(const TRK_CAS float* tab, int np, const float (&px)[NL], const float (&py)[NL],
                                                 const float (&pz)[NL], float (&bk)[NL], const Tick& tick) {
    if constexpr (J < 8) {
        if constexpr (J < TRK_OBJ_TICK_SLOTS) tick.template at<J>();
        if (J < np) {
            const trk_f2 cx = {tab[8 * J + 0], tab[8 * J + 1]}, cy = {tab[8 * J + 2], tab[8 * J + 3]},
                         cz = {tab[8 * J + 4], tab[8 * J + 5]}, cw = {tab[8 * J + 6], tab[8 * J + 7]};
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const trk_f2 key = __builtin_elementwise_fma(trk_f2{px[l], px[l]}, cx,
                                   __builtin_elementwise_fma(trk_f2{py[l], py[l]}, cy,
                                   __builtin_elementwise_fma(trk_f2{pz[l], pz[l]}, cz, cw)));
                const float ks = __uint_as_float((__float_as_uint(key.x) & ~15u) | (unsigned)(2 * J));
                const float kt = __uint_as_float((__float_as_uint(key.y) & ~15u) | (unsigned)(2 * J + 1));
                bk[l] = __builtin_fminf(bk[l], __builtin_fminf(ks, kt));
            }
        }
        scene_rank_pairs<NL, J + 1>(tab, np, px, py, pz, bk, tick);
    }
}

// `tick` is called once per trip of the sphere loop: the fused kernel uses it to trickle its link-position
// stores out between the arithmetic instead of issuing them as one burst (see spec_common: PosFlusher).
// bias corrections of the Adam iterations one launch of the IK kernel runs (passed by value: no device buffer, no copy)
#define TRK_IK_MAX_STEPS 32
struct IkSchedule { float bc1[TRK_IK_MAX_STEPS]; float rsqrt_bc2[TRK_IK_MAX_STEPS]; };

#define TRK_LDS_SPHERES 16     // sphere centres a fused kernel may keep in LDS for the arg-min gather
#define TRK_LDS_PRIMS 16       // primitive records (two float4s each) a fused kernel may keep in LDS for the winning box's gather

// Box objects of the scene for the BOX instantiations of the fused kernels (the launch chose them: boxes present, primitive table
// <= TRK_LDS_PRIMS records, copied to the wave's LDS).  Primitive-major: a record is fetched ONCE for the whole group of points (a
// scalar load per box instead of per box and point, NL independent chains per record), the loop carries only (value, index) per point
// -- compare + two selects per box and point instead of compare + eight, all half-rate instructions -- and the winner's offset and
// u = |d| - half + r are re-derived from its record in LDS.  Same strict-less, first-wins choice and the same gradient formulas as
// the loop in scene_min_sdf (which the other instantiations keep, textually untouched: the headline kernel's register allocation
// follows that text).
template <int NL>
__device__ __forceinline__ void scene_boxes_lds(const DevCostHdr& C, const float (&px)[NL], const float (&py)[NL], const float (&pz)[NL],
                                                float (&s)[NL], float (&gx)[NL], float (&gy)[NL], float (&gz)[NL], const float4* lds_prims) {
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    for (int b = 0; b < C.n_box_objects; ++b) {
        const int o = cptr(C.box_objects)[b];
        const DevObj O = load_obj(C.objects, o);
        const bool ident = (O.identity & TRK_OBJ_IDENTITY) != 0;
        float lx[NL], ly[NL], lz[NL], bv[NL];
        int bi[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float dx = px[l] - O.pos[0], dy = py[l] - O.pos[1], dz = pz[l] - O.pos[2];
            if (ident) { lx[l] = dx; ly[l] = dy; lz[l] = dz; }
            else {
                lx[l] = fmaf(O.R[0], dx, fmaf(O.R[3], dy, O.R[6] * dz));
                ly[l] = fmaf(O.R[1], dx, fmaf(O.R[4], dy, O.R[7] * dz));
                lz[l] = fmaf(O.R[2], dx, fmaf(O.R[5], dy, O.R[8] * dz));
            }
            bv[l] = __builtin_inff(); bi[l] = O.prim_begin;
        }
        for (int pi = O.prim_begin; pi < O.prim_end; ++pi) {
            const DevPrim P = load_prim(C.prims, pi);
            if (P.type == TRK_PRIM_SPHERE) continue;                     // spheres live in the merged table
            // `- half + r` folded into one constant per axis: three subtractions per primitive instead of three additions per
            // primitive and point (r == 0 for sharp boxes; the ranking value differs from (|d| - half) + r in the last place only)
            const float kx = P.hx - P.r, ky = P.hy - P.r, kz = P.hz - P.r;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const float dx = lx[l] - P.cx, dy = ly[l] - P.cy, dz = lz[l] - P.cz;
                const float ux = __builtin_fabsf(dx) - kx, uy = __builtin_fabsf(dy) - ky, uz = __builtin_fabsf(dz) - kz;
                const float mu = __builtin_fmaxf(__builtin_fmaxf(ux, uy), uz);
                float v = mu;
                if (P.type != TRK_PRIM_SHARP_BOX) {
                    const float rx = __builtin_fmaxf(ux, 0.0f), ry = __builtin_fmaxf(uy, 0.0f), rz = __builtin_fmaxf(uz, 0.0f);
                    v = __builtin_fminf(mu, 0.0f) + trk_sqrt(fmaf(rx, rx, fmaf(ry, ry, rz * rz))) - P.r;
                }
                const bool take = v < bv[l];
                bv[l] = take ? v : bv[l]; bi[l] = take ? pi : bi[l];
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (!(bv[l] < s[l])) continue;                               // this object does not beat the scene's best for this point
            lds_cfloat* rec = (lds_cfloat*)reinterpret_cast<const float*>(lds_prims) + 8 * bi[l];
            const float cx = rec[1], cy = rec[2], cz = rec[3], hx = rec[4], hy = rec[5], hz = rec[6], r = rec[7];
            const float bsharp = __float_as_int(rec[0]) == TRK_PRIM_SHARP_BOX ? 1.0f : 0.0f;
            const float bdx = lx[l] - cx, bdy = ly[l] - cy, bdz = lz[l] - cz;
            const float bux = __builtin_fabsf(bdx) - hx + r, buy = __builtin_fabsf(bdy) - hy + r, buz = __builtin_fabsf(bdz) - hz + r;
            const float bvl = bv[l];
            // gradient of the winning box in the object frame (prim_sdf's formulas): arg-max with "first maximum wins"
            int am = 0; float mu = bux;
            if (buy > mu) { mu = buy; am = 1; }
            if (buz > mu) { mu = buz; am = 2; }
            const float sx = bdx > 0.0f ? 1.0f : (bdx < 0.0f ? -1.0f : 0.0f);
            const float sy = bdy > 0.0f ? 1.0f : (bdy < 0.0f ? -1.0f : 0.0f);
            const float sz = bdz > 0.0f ? 1.0f : (bdz < 0.0f ? -1.0f : 0.0f);
            float ax, ay, az;
            if (bsharp != 0.0f) {
                ax = am == 0 ? sx : 0.0f; ay = am == 1 ? sy : 0.0f; az = am == 2 ? sz : 0.0f;
            } else {
                const float rx = __builtin_fmaxf(bux, 0.0f), ry = __builtin_fmaxf(buy, 0.0f), rz = __builtin_fmaxf(buz, 0.0f);
                const float nn = trk_sqrt(fmaf(rx, rx, fmaf(ry, ry, rz * rz)));
                const float inv = nn > 0.0f ? trk_rcp(nn) : 0.0f;
                const float inside = mu < 0.0f ? 1.0f : 0.0f;
                ax = (rx * inv + (am == 0 ? inside : 0.0f)) * sx;
                ay = (ry * inv + (am == 1 ? inside : 0.0f)) * sy;
                az = (rz * inv + (am == 2 ? inside : 0.0f)) * sz;
            }
            s[l] = bvl;
            if (ident) { gx[l] = ax; gy[l] = ay; gz[l] = az; }
            else {
                gx[l] = fmaf(O.R[0], ax, fmaf(O.R[1], ay, O.R[2] * az));
                gy[l] = fmaf(O.R[3], ax, fmaf(O.R[4], ay, O.R[5] * az));
                gz[l] = fmaf(O.R[6], ax, fmaf(O.R[7], ay, O.R[8] * az));
            }
        }
    }
}

// The scene's minimum signed distance at ONE point that is the same for every lane of the wavefront (a collision link whose position is
// a constant of the model: the Panda's first link origin sits on the base's axis) -- evaluated COOPERATIVELY instead of 64 times: lane k
// takes sphere k / primitive k of each box object, then a wave-wide minimum.  Value only: a point no joint moves has no gradient to
// hand back.  Same functions as scene_min_sdf (prim_sdf, grid_sdf); the association order of the minimum does not matter to a min.
// Box scenes spend a fifth of their primitive loop on such a link (11 - 14 boxes x 20 instructions + the winner blocks: ~300 of 2600
// vector instructions per wavefront on the Panda); here it is ~45.
// lds_prims (nullable): the wave's LDS copy of the primitive table (spec_load_prims_issue: tables up to TRK_LDS_PRIMS records) -- a
// per-lane record then comes from LDS (lgkmcnt) instead of a vector load, whose s_waitcnt vmcnt would also wait for every position
// store the wave has in flight (first version: shelf 17.59 -> 17.54 us, i.e. nothing; loads and stores share vmcnt on this ISA).
__device__ __forceinline__ float scene_min_sdf_uniform_point(const DevCostHdr& C, float x, float y, float z, int lane,
                                                             const float4* lds_prims = nullptr) {
    float v = __builtin_inff();
    for (int k = lane; k < C.n_spheres; k += TRK_WAVE) {
        const float4 S = C.spheres[k];
        const float dx = x - S.x, dy = y - S.y, dz = z - S.z;
        const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        v = __builtin_fminf(v, fmaf(n2, trk_rsq(__builtin_fmaxf(n2, 1.17549435e-38f)), -S.w));
    }
    for (int b = 0; b < C.n_box_objects; ++b) {
        const int o = cptr(C.box_objects)[b];
        const DevObj O = load_obj(C.objects, o);
        const float dx = x - O.pos[0], dy = y - O.pos[1], dz = z - O.pos[2];
        float lx = dx, ly = dy, lz = dz;
        if (!(O.identity & TRK_OBJ_IDENTITY)) {
            lx = fmaf(O.R[0], dx, fmaf(O.R[3], dy, O.R[6] * dz));
            ly = fmaf(O.R[1], dx, fmaf(O.R[4], dy, O.R[7] * dz));
            lz = fmaf(O.R[2], dx, fmaf(O.R[5], dy, O.R[8] * dz));
        }
        for (int pi = O.prim_begin + lane; pi < O.prim_end; pi += TRK_WAVE) {
            DevPrim P;
            if (lds_prims && C.n_prims <= TRK_LDS_PRIMS) {
                typedef __attribute__((address_space(3))) const float lds_cfloat;
                lds_cfloat* rec = (lds_cfloat*)reinterpret_cast<const float*>(lds_prims) + 8 * pi;
                P.type = __float_as_int(rec[0]); P.cx = rec[1]; P.cy = rec[2]; P.cz = rec[3]; P.hx = rec[4]; P.hy = rec[5]; P.hz = rec[6]; P.r = rec[7];
            } else {
                P = C.prims[pi];                                 // per-lane record: a vector load, 32 bytes
            }
            if (P.type == TRK_PRIM_SPHERE) continue;             // spheres live in the merged table
            float gx, gy, gz;
            v = __builtin_fminf(v, prim_sdf<false>(P, lx, ly, lz, gx, gy, gz));
        }
    }
#pragma unroll
    for (int m = TRK_WAVE / 2; m > 0; m >>= 1) v = __builtin_fminf(v, __shfl_xor(v, m, TRK_WAVE));
    if (C.has_grid) { float gx, gy, gz; v = __builtin_fminf(v, grid_sdf(C.grid, x, y, z, gx, gy, gz)); }   // the same cell for every lane
    return v;
}

// FAST: the caller guarantees (wave-uniformly, from the cost model header: scene_is_fast) that the scene is 1..16
// spheres of one radius and nothing else, so only that path is compiled -- a kernel that inlines this function many
// times (attached-point kernels: once per group of points) would otherwise not fit the instruction cache.
__host__ __device__ inline bool scene_is_fast(const DevCostHdr& C) {
    return C.n_spheres > 0 && C.n_spheres <= 16 && C.spheres_uniform_r && C.n_box_objects == 0 && !C.has_grid;
}

// The generated rollout kernels have two scene instantiations, chosen at the launch: SPHERES ONLY (GENERAL = false: the text below
// stops after the sphere table) and the GENERAL one (boxes -- held in LDS when the table fits TRK_LDS_PRIMS -- and / or a voxel
// grid).  Same-box A/B on the headline (profiles/r04_ab_headline_*.txt): with the brick-tiled grid index compiled in behind a
// run-time `if (C.has_grid)` the Panda sphere-scene kernel allocated 126 registers + 16 SGPR spills and took 9.55 us per launch;
// without any box / grid text 118 + 0 and 9.45 us (configuration 3: 10.5 -> 9.95 us).
__host__ __device__ inline bool scene_is_general(const DevCostHdr& C) { return C.n_box_objects > 0 || C.has_grid; }

template <int NL, class Tick = NoTick, bool FAST = false, bool GENERAL = true>
__device__ __forceinline__ void scene_min_sdf(const DevCostHdr& C, const float (&px)[NL], const float (&py)[NL],
                                              const float (&pz)[NL], float (&s)[NL], float (&gx)[NL], float (&gy)[NL],
                                              float (&gz)[NL], Tick&& tick = Tick(), const float4* lds_spheres = nullptr,
                                              const float4* lds_prims = nullptr) {
#pragma unroll
    for (int l = 0; l < NL; ++l) { s[l] = __builtin_inff(); gx[l] = 0.0f; gy[l] = 0.0f; gz[l] = 0.0f; }
    // contract with the caller: all TRK_OBJ_TICK_SLOTS tick slots are issued on every path -- interleaved with the pair ranking when
    // that path runs, in one go otherwise
    const bool paired = FAST || (C.n_spheres > 0 && C.spheres_uniform_r && C.n_spheres <= 16);
    // A voxel-grid scene gathers per lane from global memory, and on this ISA loads and stores share `vmcnt`: a gather issued
    // behind position stores waits until HBM has taken them (measured: 37 us per launch for the 200^3 grid against 11 us for the
    // analytic spheres).  So the tick slots of a grid scene fire AFTER its gathers have returned; box-only scenes read their
    // tables through the scalar cache (lgkmcnt) and keep the stores in flight under their arithmetic.
    const bool ticks_last = !FAST && GENERAL && C.has_grid;
    if (!paired && !ticks_last) {
        scene_all_ticks<0>(tick);
    }
    if (FAST || C.n_spheres > 0) {
        if (FAST || C.spheres_uniform_r) {
            // equal radii: arg-min over spheres of |p-c|^2, ranked by |p-c|^2 - |p|^2 = p.(-2c) + |c|^2 (3 FMAs per
            // sphere and point); the exact distance is recomputed for the winner only (one sqrt per point).
            int bi[NL];
            if (FAST || C.n_spheres <= 16) {
                // index rides in the 4 low mantissa bits of the ranking key: one v_and_or + half a v_min3 per sphere and
                // point.  Only near-ties (relative gap < 2^-19) can pick the other sphere, and then both distances
                // agree to ~2e-6 -- below the stated cost tolerance; the value itself is always exact.
                // Two spheres (S, T) per trip, one packed lane each: key(S), key(T) of a point are ONE v_pk_fma_f32 chain
                // over the pair record [Sx Tx | Sy Ty | Sz Tz | Sw Tw] (measured on gfx950: v_pk_fma_f32 4.4 cycles per
                // wave = 2.2 per FMA, a scalar FMA with an SGPR operand 4.2; tools/valu_microbench3.hip).
                float bk[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) bk[l] = __builtin_inff();
                // <= 8 pairs, unrolled with a wave-uniform guard each: the sphere indices are immediates, the records sit at
                // immediate offsets (first four requested together, the rest while pairs 2-3 are ranked), and the caller's
                // eight tick slots are interleaved one per pair whatever the sphere count is.
                const TRK_CAS float* tab = cptr(C.sphere_pairs);
                const int np = C.n_sphere_pairs;
                if constexpr (TickIsNoTick<Tick>::value && !TRK_EXP_UNROLLED_RANK) {
                    // nothing to interleave (table-driven kernels, attached-point kernels that evaluate the scene once
                    // per group of points): a rolled loop keeps those kernels inside the instruction cache
                    for (int j = 0; j < np; ++j) {
                        const F8 rec = load_f8_uniform(tab, j);
                        const unsigned k0 = 2u * (unsigned)j;
                        unsigned k1;                               // k0 | 1, kept opaque: else the compiler splits the
                        asm("s_or_b32 %0, %1, 1" : "=s"(k1) : "s"(k0) : "scc");   // second v_and_or into v_and + v_or3
                        const trk_f2 cx = {rec.v[0], rec.v[1]}, cy = {rec.v[2], rec.v[3]}, cz = {rec.v[4], rec.v[5]},
                                     cw = {rec.v[6], rec.v[7]};
#pragma unroll
                        for (int l = 0; l < NL; ++l) {
                            const trk_f2 key = __builtin_elementwise_fma(trk_f2{px[l], px[l]}, cx,
                                               __builtin_elementwise_fma(trk_f2{py[l], py[l]}, cy,
                                               __builtin_elementwise_fma(trk_f2{pz[l], pz[l]}, cz, cw)));
                            const float ks = __uint_as_float((__float_as_uint(key.x) & ~15u) | k0);
                            const float kt = __uint_as_float((__float_as_uint(key.y) & ~15u) | k1);
                            bk[l] = __builtin_fminf(bk[l], __builtin_fminf(ks, kt));
                        }
                    }
                } else {
                    scene_rank_pairs<NL, 0>(tab, np, px, py, pz, bk, tick);
                }
#pragma unroll
                for (int l = 0; l < NL; ++l) bi[l] = (int)(__float_as_uint(bk[l]) & 15u);
            } else {
                float bn[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) { bn[l] = __builtin_inff(); bi[l] = 0; }
                for (int k = 0; k < C.n_spheres; ++k) {
                    const F4 S = load_f4_uniform(C.spheres_sel, k);
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const float t = fmaf(px[l], S.x, fmaf(py[l], S.y, fmaf(pz[l], S.z, S.w)));
                        const bool lt = t < bn[l];
                        bi[l] = lt ? k : bi[l];
                        bn[l] = lt ? t : bn[l];
                    }
                }
            }
            // per-lane gather of the winning centre: from the wave's LDS copy when there is one (~100 cycles), else from global
            // memory (L2 hit, ~700 cycles with every wave of the chip asking at once).  Two separate loops, NOT
            // `cond ? lds[i] : global[i]`: that is one load through a selected pointer, i.e. a FLAT load, whose
            // `s_waitcnt vmcnt(0)` also waits for every output store the wave has in flight -- the position stream then
            // stops at the gather until HBM has taken all of it.
            float4 Sv[NL];
            if (lds_spheres && (FAST || C.n_spheres <= TRK_LDS_SPHERES)) {
                // an explicit LDS pointer: as two generic pointers the compiler sinks both loops' loads into one flat load again
                typedef __attribute__((address_space(3))) const float lds_cfloat;
                lds_cfloat* lp = (lds_cfloat*)reinterpret_cast<const float*>(lds_spheres);
#pragma unroll
                for (int l = 0; l < NL; ++l) Sv[l] = make_float4(lp[4 * bi[l]], lp[4 * bi[l] + 1], lp[4 * bi[l] + 2], 0.0f);
            } else {
#pragma unroll
                for (int l = 0; l < NL; ++l) Sv[l] = C.spheres[bi[l]];
            }
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const float4 S = Sv[l];
                const float dx = px[l] - S.x, dy = py[l] - S.y, dz = pz[l] - S.z;
                const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                // torch.norm backward is 0 at p == c: with n2 floored, d * inv = 0 * 1.8e19 = 0 there (and n2 * inv = 0)
                const float inv = trk_rsq(__builtin_fmaxf(n2, 1.17549435e-38f));     // one transcendental per point
                s[l] = fmaf(n2, inv, -C.sphere_r); gx[l] = dx * inv; gy[l] = dy * inv; gz[l] = dz * inv;
            }
        } else {
            for (int k = 0; k < C.n_spheres; ++k) {
                const F4 S = load_f4_uniform(C.spheres, k);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const float dx = px[l] - S.x, dy = py[l] - S.y, dz = pz[l] - S.z;
                    const float nrm = trk_sqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
                    const float v = nrm - S.w;
                    const bool take = v < s[l];
                    const float inv = nrm > 0.0f ? trk_rcp(nrm) : 0.0f;
                    s[l] = take ? v : s[l];
                    gx[l] = take ? dx * inv : gx[l]; gy[l] = take ? dy * inv : gy[l]; gz[l] = take ? dz * inv : gz[l];
                }
            }
        }
    }
    if (FAST || !GENERAL) return;
    // objects that still have non-sphere primitives (boxes): empty loop for sphere-only scenes.
    // Per object and point, the primitive loop computes VALUES only and carries the winner's offset d and u = |d| - half + r along
    // (six selects instead of a gradient per primitive); the gradient is formed once, for the winner (primitives.py:327-334:
    // only the arg-min primitive passes a gradient).  Point by point on purpose: the winner's eight registers are live across
    // the primitive loop, and with all NL points in flight at once the Panda kernel no longer fitted its 128 registers (57
    // spills in the sphere-scene hot path as well: 10 -> 19.7 us).
    if (lds_prims && C.n_prims <= TRK_LDS_PRIMS) {
        scene_boxes_lds<NL>(C, px, py, pz, s, gx, gy, gz, lds_prims);       // BOX instantiations: primitive-major, (value, index)
    } else
    for (int b = 0; b < C.n_box_objects; ++b) {
        const int o = cptr(C.box_objects)[b];
        const DevObj O = load_obj(C.objects, o);
        const bool ident = (O.identity & TRK_OBJ_IDENTITY) != 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            float lx, ly, lz;
            {
                const float dx = px[l] - O.pos[0], dy = py[l] - O.pos[1], dz = pz[l] - O.pos[2];
                if (ident) { lx = dx; ly = dy; lz = dz; }
                else {
                    lx = fmaf(O.R[0], dx, fmaf(O.R[3], dy, O.R[6] * dz));
                    ly = fmaf(O.R[1], dx, fmaf(O.R[4], dy, O.R[7] * dz));
                    lz = fmaf(O.R[2], dx, fmaf(O.R[5], dy, O.R[8] * dz));
                }
            }
            float bv = __builtin_inff(), bdx = 0.0f, bdy = 0.0f, bdz = 0.0f, bux = 0.0f, buy = 0.0f, buz = 0.0f, bsharp = 0.0f;
            for (int pi = O.prim_begin; pi < O.prim_end; ++pi) {
                const DevPrim P = load_prim(C.prims, pi);
                if (P.type == TRK_PRIM_SPHERE) continue;                     // spheres live in the merged table above
                const bool sharp = P.type == TRK_PRIM_SHARP_BOX;             // wave-uniform; P.r == 0 for sharp boxes
                const float dx = lx - P.cx, dy = ly - P.cy, dz = lz - P.cz;
                const float ux = __builtin_fabsf(dx) - P.hx + P.r, uy = __builtin_fabsf(dy) - P.hy + P.r, uz = __builtin_fabsf(dz) - P.hz + P.r;
                const float mu = __builtin_fmaxf(__builtin_fmaxf(ux, uy), uz);
                float v = mu;
                if (!sharp) {
                    const float rx = __builtin_fmaxf(ux, 0.0f), ry = __builtin_fmaxf(uy, 0.0f), rz = __builtin_fmaxf(uz, 0.0f);
                    v = __builtin_fminf(mu, 0.0f) + trk_sqrt(fmaf(rx, rx, fmaf(ry, ry, rz * rz))) - P.r;
                }
                const bool take = v < bv;
                bv = take ? v : bv;
                bdx = take ? dx : bdx; bdy = take ? dy : bdy; bdz = take ? dz : bdz;
                bux = take ? ux : bux; buy = take ? uy : buy; buz = take ? uz : buz;
                bsharp = take ? (sharp ? 1.0f : 0.0f) : bsharp;
            }
            if (!(bv < s[l])) continue;                                   // this object does not beat the scene's best for this point
            // gradient of the winning box in the object frame (prim_sdf's formulas): arg-max with "first maximum wins"
            int am = 0; float mu = bux;
            if (buy > mu) { mu = buy; am = 1; }
            if (buz > mu) { mu = buz; am = 2; }
            const float sx = bdx > 0.0f ? 1.0f : (bdx < 0.0f ? -1.0f : 0.0f);
            const float sy = bdy > 0.0f ? 1.0f : (bdy < 0.0f ? -1.0f : 0.0f);
            const float sz = bdz > 0.0f ? 1.0f : (bdz < 0.0f ? -1.0f : 0.0f);
            float ax, ay, az;
            if (bsharp != 0.0f) {
                ax = am == 0 ? sx : 0.0f; ay = am == 1 ? sy : 0.0f; az = am == 2 ? sz : 0.0f;
            } else {
                const float rx = __builtin_fmaxf(bux, 0.0f), ry = __builtin_fmaxf(buy, 0.0f), rz = __builtin_fmaxf(buz, 0.0f);
                const float nn = trk_sqrt(fmaf(rx, rx, fmaf(ry, ry, rz * rz)));
                const float inv = nn > 0.0f ? trk_rcp(nn) : 0.0f;
                const float inside = mu < 0.0f ? 1.0f : 0.0f;
                ax = (rx * inv + (am == 0 ? inside : 0.0f)) * sx;
                ay = (ry * inv + (am == 1 ? inside : 0.0f)) * sy;
                az = (rz * inv + (am == 2 ? inside : 0.0f)) * sz;
            }
            s[l] = bv;
            if (ident) { gx[l] = ax; gy[l] = ay; gz[l] = az; }
            else {
                gx[l] = fmaf(O.R[0], ax, fmaf(O.R[1], ay, O.R[2] * az));
                gy[l] = fmaf(O.R[3], ax, fmaf(O.R[4], ay, O.R[5] * az));
                gz[l] = fmaf(O.R[6], ax, fmaf(O.R[7], ay, O.R[8] * az));
            }
        }
    }
    // precomputed voxel grid (kept out of the loop above: its index arithmetic is loop-invariant and the
    // compiler would otherwise hoist 15 IEEE divisions in front of every scene, grid or not)
    if (C.has_grid) {
        // all NL cells are addressed first and their 4 + 12 bytes requested together: one memory round trip for the group
        int64_t lin[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) lin[l] = grid_cell(C.grid, px[l], py[l], pz[l]);
        float4 cell[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) cell[l] = C.grid.cells[lin[l]];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const bool take = cell[l].w < s[l];
            s[l] = take ? cell[l].w : s[l]; gx[l] = take ? cell[l].x : gx[l]; gy[l] = take ? cell[l].y : gy[l]; gz[l] = take ? cell[l].z : gz[l];
        }
        if (!paired) scene_all_ticks<0>(tick);
    }
}

// Geometric Jacobian read-out (robot_tree.py:230-246): the walk has left, per lane, one record (z, p) per contributing joint and
// the target link's position in `rec` (lane-major, odd stride `rstride`: conflict-free both ways); `slot[d]` = record of DOF d or
// -1 (column stays zero).  lin_jac / ang_jac [N,3,D] leave as the wave's contiguous run of 64 * 3D floats: four consecutive
// elements per lane and one 16-byte write-through store per array.
__device__ __forceinline__ void trk_jac_readout(float* rec, const int* slot, int rstride, int n_cols, int D, int rows,
                                                float* lo, float* ao, int lane) {
    const int w3 = 3 * D;
    {   // each lane first turns its own records (z, p_joint) into (z, z x (p_link - p_joint)): the transposed read-out below
        // is then one LDS read per output element instead of five reads and a cross-product component
        float* mine = rec + lane * rstride;
        const float e0 = mine[6 * n_cols], e1 = mine[6 * n_cols + 1], e2 = mine[6 * n_cols + 2];
        for (int c = 0; c < n_cols; ++c) {
            float* j = mine + 6 * c;
            const float z0 = j[0], z1 = j[1], z2 = j[2], r0 = e0 - j[3], r1 = e1 - j[4], r2 = e2 - j[5];
            j[3] = z1 * r2 - z2 * r1; j[4] = z2 * r0 - z0 * r2; j[5] = z0 * r1 - z1 * r0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    auto element = [&](int sl, int rem, float& lv, float& av) {
        const int r = (rem >= D) + (rem >= 2 * D), d = rem - r * D;
        const int c = slot[d];
        lv = 0.0f; av = 0.0f;
        if (c >= 0) {
            const float* j = rec + sl * rstride + 6 * c;
            av = j[r]; lv = j[3 + r];
        }
    };
    if (rows == TRK_WAVE && ((reinterpret_cast<uintptr_t>(lo) | reinterpret_cast<uintptr_t>(ao)) & 15) == 0) {
        const float inv_w3 = 1.0f / (float)w3;
        for (int e = lane; e < 16 * w3; e += TRK_WAVE) {              // 64 * 3D floats = 16 * 3D float4 per array
            const int k0 = 4 * e;
            int sl = (int)((float)k0 * inv_w3);
            sl -= (sl * w3 > k0); sl += ((sl + 1) * w3 <= k0);
            int rem = k0 - sl * w3;
            float lv[4], av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                element(sl, rem, lv[j], av[j]);
                if (++rem == w3) { rem = 0; ++sl; }
            }
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 l4 = {lv[0], lv[1], lv[2], lv[3]}, a4 = {av[0], av[1], av[2], av[3]};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(lo + k0), "v"(l4) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(ao + k0), "v"(a4) : "memory");
        }
    } else {
        int sl = lane / w3, rem = lane - sl * w3;
        const int ds = TRK_WAVE / w3, dr = TRK_WAVE - ds * w3;
        const int64_t count = (int64_t)rows * w3;
        for (int64_t k = lane; k < count; k += TRK_WAVE) {
            float lv, av;
            element(sl, rem, lv, av);
            lo[k] = lv; ao[k] = av;
            sl += ds; rem += dr;
            if (rem >= w3) { rem -= w3; ++sl; }
        }
    }
}

// Frame.get_quaternion (trace method with M[3][3] = 1, frame.py:87-114), then xyzw -> wxyz (quaternion.py:240-242)
__device__ __forceinline__ void frame_quat_wxyz(const float* R, float* out) {
    float t = R[0] + R[4] + R[8] + 1.0f;
    float qx, qy, qz, qw;
    if (t > 1.0f) {
        qw = t; qz = R[3] - R[1]; qy = R[2] - R[6]; qx = R[7] - R[5];
    } else {
        // i = arg-max diagonal with the reference's comparison order; (i, j, k) cyclic
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
        if (i == 0) {
            t = R[0] - (R[4] + R[8]) + 1.0f;
            qx = t; qy = R[1] + R[3]; qz = R[6] + R[2]; qw = R[7] - R[5];
        } else if (i == 1) {
            t = R[4] - (R[8] + R[0]) + 1.0f;
            qy = t; qz = R[5] + R[7]; qx = R[1] + R[3]; qw = R[2] - R[6];
        } else {
            t = R[8] - (R[0] + R[4]) + 1.0f;
            qz = t; qx = R[6] + R[2]; qy = R[5] + R[7]; qw = R[3] - R[1];
        }
    }
    const float sc = 0.5f / sqrtf(t);
    out[0] = qw * sc; out[1] = qx * sc; out[2] = qy * sc; out[3] = qz * sc;
}

// Rotation vector of a rotation matrix Re (row-major 3x3) for the Gauss-Newton IK residual (BUILD-DEFINED, oracle_impl.inc
// orc_ik_gn_step): quaternion (w, v) by Frame.get_quaternion's trace method, sign chosen so that w >= 0, then
// v / |v| * 2 atan2(|v|, w) -- well conditioned up to a rotation of pi, where |v| -> 1.
__device__ __forceinline__ void trk_rotvec(const float* Re, float* out) {
    float qe[4];
    frame_quat_wxyz(Re, qe);
    const float sgn = qe[0] < 0.0f ? -1.0f : 1.0f;
    const float w = __builtin_fminf(qe[0] * sgn, 1.0f);
    const float vx = qe[1] * sgn, vy = qe[2] * sgn, vz = qe[3] * sgn;
    const float nv = sqrtf(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
    const float sc = 2.0f * atan2f(nv, w) / __builtin_fmaxf(nv, 1e-12f);
    out[0] = vx * sc; out[1] = vy * sc; out[2] = vz * sc;
}

// (A + 0) x = g for a symmetric positive definite D x D matrix held by ONE lane in registers: A = the lower triangle, packed
// row-major (A[i (i + 1) / 2 + j], j <= i), overwritten by its Cholesky factor; g is overwritten by the solution.  Every loop has
// compile-time bounds, so for D <= 9 the whole factorisation is straight-line register code (D = 7: 28 + 7 registers).
template <int D>
__device__ __forceinline__ void trk_chol_solve(float (&A)[D * (D + 1) / 2], float (&g)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
        float d = A[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fmaf(-A[j * (j + 1) / 2 + k], A[j * (j + 1) / 2 + k], d);
        const float inv = trk_rsq(__builtin_fmaxf(d, 1e-30f));      // 1 / L_jj
        A[j * (j + 1) / 2 + j] = inv;                                // the diagonal keeps the RECIPROCAL: no division below
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            float a = A[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) a = fmaf(-A[i * (i + 1) / 2 + k], A[j * (j + 1) / 2 + k], a);
            A[i * (i + 1) / 2 + j] = a * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {                                    // L y = g
        float a = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) a = fmaf(-A[i * (i + 1) / 2 + k], g[k], a);
        g[i] = a * A[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {                               // L^T x = y
        float a = g[i];
#pragma unroll
        for (int k = i + 1; k < D; ++k) a = fmaf(-A[k * (k + 1) / 2 + i], g[k], a);
        g[i] = a * A[i * (i + 1) / 2 + i];
    }
}

// workspace box, distance_fields.py:326-332: max_k (margin - sd_k) over the six planes; returns the value,
// adds scale * d/dp to (ax, ay, az)
__device__ __forceinline__ float ws_cost_point(const DevCostHdr& C, float mg, float x, float y, float z, float scale,
                                               float& ax, float& ay, float& az) {
    const bool clamp = (C.clamp_fields & TRK_FIELD_WS) != 0;         // wave-uniform
    // the six signed plane distances are {p_k - min_k, max_k - p_k}; per axis the smaller one is h_k - |p_k - c_k|,
    // so max_planes (margin - sd) = margin - min_k (h_k - |d_k|) and the gradient is sign(d_a) on the arg-min axis
    // (20 instructions instead of 35; differs from the plane form by fp32 rounding of c and h, ~1e-7)
    const float dx = x - C.ws_c[0], dy = y - C.ws_c[1], dz = z - C.ws_c[2];
    const float mx = C.ws_h[0] - __builtin_fabsf(dx), my = C.ws_h[1] - __builtin_fabsf(dy), mz = C.ws_h[2] - __builtin_fabsf(dz);
    const float m = __builtin_fminf(__builtin_fminf(mx, my), mz);
    const bool e0 = mx == m, e1 = (my == m) && !e0, e2 = !(e0 || e1);          // first arg-min axis wins
    const float v = mg - m;
    if (clamp) scale = v > 0.0f ? scale : 0.0f;                      // relu: no gradient at or below zero
    ax += e0 ? __builtin_copysignf(scale, dx) : 0.0f;
    ay += e1 ? __builtin_copysignf(scale, dy) : 0.0f;
    az += e2 ? __builtin_copysignf(scale, dz) : 0.0f;
    return clamp ? __builtin_fmaxf(v, 0.0f) : v;
}

// EE SE(3) tracking, distance_fields.py:347-356 + geometrics/utils.py:148-154.
// Returns the cost; gR[9], gt[3] receive d cost / d (R, t).
__device__ __forceinline__ float ee_cost_eval(const float* R, const float* t, const float* Ht /*16*/,
                                              float w_pos, float w_rot, int square, float* gR, float* gt) {
    float tr = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) tr = fmaf(R[3 * r + c], Ht[4 * r + c], tr);
    const float dx = t[0] - Ht[3], dy = t[1] - Ht[7], dz = t[2] - Ht[11];
    const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rs = n2 > 0.0f ? trk_rsq(n2) : 0.0f;
    const float nrm = n2 * rs;
    float d = 0.0f;
    if (w_rot > 0.0f) d = fmaf(w_rot, 1.0f - (tr - 1.0f) * 0.5f, d);
    if (w_pos > 0.0f) d = fmaf(w_pos, nrm, d);
    const float sc = square ? 2.0f * d : 1.0f;
    const float kr = w_rot > 0.0f ? -0.5f * w_rot * sc : 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) gR[3 * r + c] = kr * Ht[4 * r + c];
    const float kp = w_pos > 0.0f ? sc * w_pos * rs : 0.0f;
    gt[0] = kp * dx; gt[1] = kp * dy; gt[2] = kp * dz;
    return square ? d * d : d;
}

// d quat (wxyz) of rotation_matrix_to_q (quaternion.py:135-166) for a rotation m (row-major 9) moving by dm: the derivative of the
// selected candidate (shared by the table-driven analytic-Jacobian kernel and the generated one)
__device__ __forceinline__ void quat_jvp(const float* m, const float* dm, float* dq) {
    const float a[4] = {1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8]};
    const float da[4] = {dm[0] + dm[4] + dm[8], dm[0] - dm[4] - dm[8], -dm[0] + dm[4] - dm[8], -dm[0] - dm[4] + dm[8]};
    float qa[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) qa[k] = a[k] > 0.0f ? sqrtf(a[k]) : 0.0f;
    int b = 0; float qb = qa[0], ab = a[0], dab = da[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if (qa[k] > qb) { qb = qa[k]; ab = a[k]; dab = da[k]; b = k; }
    const bool posa = ab > 0.0f;
    ab = posa ? ab : 0.0f; dab = posa ? dab : 0.0f;
    float N[4], dN[4];
    if (b == 0)      { N[0] = ab; N[1] = m[7] - m[5]; N[2] = m[2] - m[6]; N[3] = m[3] - m[1];
                       dN[0] = dab; dN[1] = dm[7] - dm[5]; dN[2] = dm[2] - dm[6]; dN[3] = dm[3] - dm[1]; }
    else if (b == 1) { N[0] = m[7] - m[5]; N[1] = ab; N[2] = m[3] + m[1]; N[3] = m[2] + m[6];
                       dN[0] = dm[7] - dm[5]; dN[1] = dab; dN[2] = dm[3] + dm[1]; dN[3] = dm[2] + dm[6]; }
    else if (b == 2) { N[0] = m[2] - m[6]; N[1] = m[3] + m[1]; N[2] = ab; N[3] = m[5] + m[7];
                       dN[0] = dm[2] - dm[6]; dN[1] = dm[3] + dm[1]; dN[2] = dab; dN[3] = dm[5] + dm[7]; }
    else             { N[0] = m[3] - m[1]; N[1] = m[6] + m[2]; N[2] = m[7] + m[5]; N[3] = ab;
                       dN[0] = dm[3] - dm[1]; dN[1] = dm[6] + dm[2]; dN[2] = dm[7] + dm[5]; dN[3] = dab; }
    const float den = 2.0f * fmaxf(qb, 0.1f);
    const float dden = (qb > 0.1f && posa) ? dab / qb : 0.0f;
    const float inv = 1.0f / den;
#pragma unroll
    for (int k = 0; k < 4; ++k) dq[k] = dN[k] * inv - N[k] * dden * inv * inv;
}

// quat_jvp factored for callers that differentiate ONE rotation along many directions (the generated analytic-Jacobian kernel: every
// joint of a link's chain): the candidate selection of rotation_matrix_to_q depends on the rotation only.  sqrt is monotone, so the
// candidate with the largest a_k IS the one with the largest sqrt(a_k) (first wins on ties, as there): one square root instead of four.
struct QuatSel { int b; float N[4]; float inv, kden, pos; };      // dq = dN * inv - N * (dab * kden)
__device__ __forceinline__ QuatSel quat_sel(const float* m) {
    const float a[4] = {1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8]};
    int b = 0; float ab = a[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) if (a[k] > ab && a[k] > 0.0f) { ab = a[k]; b = k; }
    const bool posa = ab > 0.0f;
    ab = posa ? ab : 0.0f;
    const float qb = sqrtf(ab);
    QuatSel s;
    s.b = b;
    if (b == 0)      { s.N[0] = ab; s.N[1] = m[7] - m[5]; s.N[2] = m[2] - m[6]; s.N[3] = m[3] - m[1]; }
    else if (b == 1) { s.N[0] = m[7] - m[5]; s.N[1] = ab; s.N[2] = m[3] + m[1]; s.N[3] = m[2] + m[6]; }
    else if (b == 2) { s.N[0] = m[2] - m[6]; s.N[1] = m[3] + m[1]; s.N[2] = ab; s.N[3] = m[5] + m[7]; }
    else             { s.N[0] = m[3] - m[1]; s.N[1] = m[6] + m[2]; s.N[2] = m[7] + m[5]; s.N[3] = ab; }
    const float den = 2.0f * fmaxf(qb, 0.1f);
    s.inv = 1.0f / den;
    s.kden = (qb > 0.1f && posa) ? s.inv * s.inv / qb : 0.0f;
    s.pos = posa ? 1.0f : 0.0f;
    return s;
}
__device__ __forceinline__ void quat_jvp_sel(const QuatSel& s, const float* dm, float* dq) {
    float dN[4];
    float dab;
    if (s.b == 0)      { dab = dm[0] + dm[4] + dm[8];  dN[0] = dab; dN[1] = dm[7] - dm[5]; dN[2] = dm[2] - dm[6]; dN[3] = dm[3] - dm[1]; }
    else if (s.b == 1) { dab = dm[0] - dm[4] - dm[8];  dN[0] = dm[7] - dm[5]; dN[1] = dab; dN[2] = dm[3] + dm[1]; dN[3] = dm[2] + dm[6]; }
    else if (s.b == 2) { dab = -dm[0] + dm[4] - dm[8]; dN[0] = dm[2] - dm[6]; dN[1] = dm[3] + dm[1]; dN[2] = dab; dN[3] = dm[5] + dm[7]; }
    else               { dab = -dm[0] - dm[4] + dm[8]; dN[0] = dm[3] - dm[1]; dN[1] = dm[6] + dm[2]; dN[2] = dm[7] + dm[5]; dN[3] = dab; }
    dN[s.b] *= s.pos;                           // (the selected a_k is positive for every rotation; the reference's guard is kept)
    const float kd = dab * s.kden;
#pragma unroll
    for (int k = 0; k < 4; ++k) dq[k] = dN[k] * s.inv - s.N[k] * kd;
}
