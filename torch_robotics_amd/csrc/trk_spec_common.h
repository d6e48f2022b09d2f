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
#ifndef __HIPCC_RTC__
#include <cstdlib>
#endif

template <bool B, class T, class F> struct TrkIf { typedef T type; };        // std::conditional (no <type_traits> under hipRTC)
template <class T, class F> struct TrkIf<false, T, F> { typedef F type; };
template <class A, class B> struct TrkSame { static constexpr bool value = false; };
template <class A> struct TrkSame<A, A> { static constexpr bool value = true; };

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
    int32_t io_f16;               // TRK_IO_*: 0 all fp32; 1 q / link_pos / gq fp16 in HBM; 2 q / link_pos fp16, gq fp32 (arithmetic, cost, cost_sum: fp32)
    unsigned long long* stamps;   // profiling hook (nullable): [n_waves][8] s_memtime stamps at phase boundaries
    // boolean mode (trk_rollout_collision): when coll_out != nullptr the kernel stops after FK, ORs the selected fields'
    // "signed distance < margin" tests and writes one byte per sample -- no positions, cost or gradient leave the chip
    uint8_t* coll_out;
    int32_t coll_fields;          // TRK_FIELD_* mask
    int32_t coll_use_default;     // 1: per-link margins + cutoff of the cost model, 0: coll_margin for every test
    float coll_margin;
    int32_t jac_stream;           // fused rollout + Jacobian (launch_rjac): 1 = the Jacobian tiles leave as non-temporal stores (the launch's
                                  // working set exceeds the Infinity Cache: spec_stream_bytes), 0 = write-through like every in-cache output
    // geometric-Jacobian kernel (trk_fk_jacobian, launch_jac): target link, index of its parent joint in the file's joint
    // list (the reference's column rule), outputs pos [N,3], quat_wxyz [N,4], lin_jac / ang_jac [N,3,D]
    int32_t jac_link, jac_joint_idx;
    int32_t jac_p_end, jac_n_cols;    // the walk may stop after this pre-order position; number of joints that get a column
    int8_t jac_slot[TRK_MAX_DOFS];    // DOF -> record slot of its column (-1: the column stays zero), in walk order
    float* jac_pos; float* jac_quat; float* jac_lin; float* jac_ang;
    // field kernel on given link positions (trk_cost_fields, launch_fields): positions [N, L, 3] in, per-sample upstream gradient
    // (nullable), cost via `cost`, position gradient [N, L, 3] out (nullable); the fields are selected through `w`
    const float* fld_pos; const float* fld_gcost; float* fld_g;
    // all-links FK kernel (trk_fk_forward with every link selected, launch_fkh): H [N, L, 4, 4] out;
    // its reverse mode (launch_fkhbwd): the adjoint gH [N, L, 4, 4] in (read only), gq out
    float* fk_H;
    // via-point mode of the boolean kernel (trk_rollout_collision_via; get_trajs_collision_and_free tasks.py:244-251): q is then a
    // batch of trajectories x [T, via_H, via_S] whose first D columns are joint positions, and sample n = (t, i, a) is the
    // configuration x[t, i] * via_alpha[a] + x[t, i + 1] * via_beta[a] (interpolate_traj_via_points trajectory/utils.py:37-50)
    const float* via_alpha; const float* via_beta;   // DEVICE [via_n]
    int32_t via_n;                // interpolated points per segment; 0 = q holds the configurations themselves
    int32_t via_H, via_S;         // way points per trajectory, floats per way point (>= D)
    // round 6 (trk_rollout_collision_via_flags): the per-trajectory flags of trk_traj_validate folded into the via-point launch --
    // bit 0: some interpolated configuration of the trajectory collides, bit 1: some joint position of its way points lies outside
    // [via_qmin, via_qmax] (NaN = outside).  A trajectory of hi samples is covered by at most K = hi / 64 + 2 wavefronts; wavefront w
    // writes what IT has seen of trajectory t to via_partial[t K + (w - floor(t hi / 64))] -- plain stores, every byte a reader looks at
    // is written on every launch: no atomics, nothing to zero.  k_traj_partition ORs a trajectory's bytes (one short row).
    uint8_t* via_partial;         // nullable
    int32_t via_slots; int32_t _pad_via;      // trajectories 64 consecutive samples can touch (trk_via_slots)
    const float* via_qmin; const float* via_qmax;    // DEVICE [D]
    // fp16 q (io_f16 != 0): the gradient is multiplied by grad_scale (fp32) before it is stored, an fp16 store saturates at
    // +-65504 instead of writing inf ("loss scaling": config 5's GP term reaches 1e5 .. 1e6 at sigma_gp = 0.1, dt = 5/128)
    float grad_scale;
    // GP-prior fused into the rollout (launch_gp: trk_rollout_gp_cost_grad): velocities in, their gradient out, the prior's
    // parameters a = 12 / (sigma^2 dt^3), b = -6 / (sigma^2 dt^2), c = 4 / (sigma^2 dt), its weight and the horizon (a lane's time
    // step is sample % gp_H; samples of one trajectory are consecutive)
    const void* qd; void* gqd;
    float gp_dt, gp_a, gp_b, gp_c, gp_w;
    int32_t gp_H;
};

#ifndef __HIPCC_RTC__          // host side of a unit (launchers, registry): not part of an in-process device compilation
struct SpecEntry;
// `self`: the entry the function was taken from -- the generated launchers ignore it, the generic launchers of a unit loaded as a
// code object (hipRTC fall-back, trk_spec_register_module) find their kernels through it
typedef void (*SpecLaunchFn)(const SpecEntry* self, const SpecArgs& args, int base_identity, hipStream_t stream);
#endif

// Arguments of the generated IK kernel (trk_ik_steps on the unit's tracked link).  A struct of its own: the 256-byte schedule
// would otherwise ride in the kernarg segment of every kernel of the unit.
struct IkArgs {
    float base_R[9];
    float base_t[3];
    const float* H_target;        // DEVICE [16] or [N,16]
    int32_t per_sample;
    int32_t n_steps;              // <= TRK_IK_MAX_STEPS
    const float* lower;           // DEVICE [D]: the (shrunk) limits of the hinge / validity test
    const float* upper;
    float w_jl, se3_eps, lr;
    int32_t _pad;
    IkSchedule sched;
    int64_t n;
    float* q; float* adam_m; float* adam_v;     // [N,D], in place
    float* loss; uint8_t* valid;                // nullable; q as passed in
};
#ifndef __HIPCC_RTC__
typedef void (*SpecIkLaunchFn)(const SpecEntry* self, const IkArgs& args, int base_identity, hipStream_t stream);
#endif

// Arguments of the generated Gauss-Newton / Levenberg-Marquardt IK kernel (trk_ik_gn_steps on the unit's tracked link): per iteration
// stateful FK + geometric Jacobian (robot_tree.py:218-248), pose residual, J^T J + lambda I, Cholesky, step, clamp -- per lane, in registers.
struct IkGnArgs {
    float base_R[9];
    float base_t[3];
    const float* H_target;        // DEVICE [16] or [N,16]
    int32_t per_sample;
    int32_t n_steps;
    const float* lower;           // DEVICE [D]: the step is clamped to [lower, upper]; also the validity test
    const float* upper;
    float damping, lm_gain, step_scale, se3_eps;
    int64_t n;
    float* q;                     // [N,D], in place
    float* err; uint8_t* valid;   // nullable; SE3 distance / validity of q as passed in
};
#ifndef __HIPCC_RTC__
typedef void (*SpecIkGnLaunchFn)(const SpecEntry* self, const IkGnArgs& args, int base_identity, hipStream_t stream);
#endif

// Layout version of SpecArgs / SpecEntry / DevCostHdr as seen by a generated unit.  A unit compiled against another layout
// (a stale on-disk JIT object) must never be dispatched: trk_spec_register refuses it.  Bump on ANY change to these structs,
// to TrkRolloutWeights or to the TRK_MAX_* limits in include/trk.h.
#define TRK_SPEC_ABI_VERSION (TRK_ABI_VERSION * 1000 + 25)

#ifndef __HIPCC_RTC__
struct SpecEntry {
    int32_t spec_abi_version;   // TRK_SPEC_ABI_VERSION the unit was compiled with
    uint32_t sizeof_args;       // sizeof(SpecArgs) + sizeof(IkArgs) the unit was compiled with
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
    SpecLaunchFn launch_jac;    // stateful FK + geometric Jacobian of one link (robot_tree.py:136-248); nullptr if not generated
    SpecLaunchFn launch_coll;   // FK + boolean collision fields (trk_rollout_collision); nullptr if not generated
    SpecLaunchFn launch_fkh;    // FK matrices of all links (trk_fk_forward, every link selected); nullptr if not generated
    SpecLaunchFn launch_fkhbwd; // its reverse mode (trk_fk_backward, every link selected): fk_H = gH in, gq out; nullptr if not generated
    SpecIkLaunchFn launch_ik;   // Adam IK iterations on ee_link, configurations and optimiser state in registers; nullptr if not generated
    SpecLaunchFn launch_fk1;    // FK matrix of one link (jac_link, jac_p_end) -> fk_H [N,4,4]; nullptr if not generated
    SpecLaunchFn launch_fields; // collision fields on given link positions (trk_cost_fields); nullptr if not generated
    // interpolated link points baked into the unit (interpolate_link_pos): column n_links + v = w[2v] * link src[2v] +
    // w[2v+1] * link src[2v+1]; obj_link_idx / self_pairs may name them.  A cost model matches only with the same table.
    int32_t n_virtual;
    const int32_t* virtual_src; // [2 * n_virtual]
    const float* virtual_w;     // [2 * n_virtual]
    SpecIkGnLaunchFn launch_ikgn;   // Gauss-Newton IK iterations on ee_link (trk_ik_gn_steps); nullptr if not generated
    // fused rollout + GP prior in one launch (trk_rollout_gp_cost_grad): returns 0, or 1 when this unit cannot serve the call
    // (self-collision pairs between independently scheduled subtrees with w_self != 0) -- the caller then runs the two-launch form
    int (*launch_gp)(const SpecEntry* self, const SpecArgs& args, int base_identity, hipStream_t stream);
    void* module_ctx;           // nullptr for a linked / dlopen-ed unit; the code-object unit's kernel table otherwise
    // fused rollout + geometric Jacobian of the unit's tracked link in one launch (trk_rollout_jacobian_cost_grad; many-link units whose
    // stateful and stateless walks coincide on the columns' chains): returns 0, or 1 when this unit does not serve the call; nullptr if
    // not generated
    int (*launch_rjac)(const SpecEntry* self, const SpecArgs& args, int base_identity, hipStream_t stream);
    // analytic Jacobian of every link (trk_fk_analytic_jacobian: d [pos, quat] / d q, [N, L, 7, D] -> args.jac_lin); nullptr if not generated
    SpecLaunchFn launch_ajac;
};

// Does this launch take the F32Stream instantiation (non-temporal output stores)?  Its working set -- q in, positions, cost and
// gradient out -- exceeds the Infinity Cache.  TRK_STREAM_STORES=0 / 1 forces the answer, TRK_STREAM_STORE_BYTES moves the threshold
// (default 256 MiB; measured: 201 MB per launch is faster write-through, 403 MB 27 % faster non-temporal).
inline bool spec_stream_bytes(double bytes) {
    // read per launch (two getenv calls, ~0.1 us): tests and A/B runs flip the switch inside one process
    if (const char* e = std::getenv("TRK_STREAM_STORES")) return std::atoi(e) != 0;
    const char* t = std::getenv("TRK_STREAM_STORE_BYTES");
    return bytes > (t ? std::atof(t) : 256.0 * 1024 * 1024);
}
inline bool spec_stream_stores(const SpecArgs& a, int n_links, int n_dofs) {
    return spec_stream_bytes((double)a.n * (4.0 * n_dofs * 2 + 4.0 + (a.link_pos ? 12.0 * n_links : 0.0)));
}

// registry filled by static initialisers of the generated translation units
// returns 0 when the unit was accepted, TRK_ERR_INVALID_ARG (and registers nothing) when its layout stamp differs
int trk_spec_register(const SpecEntry* e);
#define SPEC_ENTRY_STAMP TRK_SPEC_ABI_VERSION, (uint32_t)(sizeof(SpecArgs) + sizeof(IkArgs) + sizeof(IkGnArgs)), (uint32_t)sizeof(SpecEntry), (uint32_t)sizeof(DevCostHdr)
const SpecEntry* trk_spec_find(uint64_t model_hash, int n_links, int n_dofs);
const SpecEntry* trk_spec_find_points(uint64_t model_hash, uint64_t points_hash, int n_points);
#endif      // !__HIPCC_RTC__

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
// The output stores are `asm volatile` (never dropped, never reordered among themselves); they write memory the kernel never
// reads back, so they carry NO "memory" clobber: with one, every store is a full compiler barrier that chops the arithmetic
// between two ticks into separate scheduling regions.
#ifndef TRK_STORE_CLOBBER
#define TRK_STORE_CLOBBER
#endif
typedef float trk_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt_f4(float4* p, const float4& v) {
    const trk_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(x)  TRK_STORE_CLOBBER);
}
typedef float trk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_wt_f2(float* p, float a, float b) {
    const trk_f2 x = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(x)  TRK_STORE_CLOBBER);
}
__device__ __forceinline__ void store_wt_f1(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(v)  TRK_STORE_CLOBBER);
}

// HBM-side element type of q / link_pos / gq: four consecutive elements <-> one float4 of LDS (fp32 arithmetic always).
typedef _Float16 trk_h4 __attribute__((ext_vector_type(4)));
// The I/O mode a rollout kernel is instantiated for: Q = type of q and link_pos, G = type of the gradient, kScaled = the gradient
// is multiplied by SpecArgs::grad_scale and stored saturating.  `float` / `_Float16` name themselves; HalfG32 is the mixed mode
// (fp16 trajectories and positions, fp32 gradient).  The fp32 instantiation has no scale: its code is what it was.
struct HalfG32 {};
template <class IO> struct IoTraits { typedef IO Q; typedef IO G; static constexpr bool kScaled = false; };
template <> struct IoTraits<_Float16> { typedef _Float16 Q; typedef _Float16 G; static constexpr bool kScaled = true; };
template <> struct IoTraits<HalfG32> { typedef _Float16 Q; typedef float G; static constexpr bool kScaled = true; };
// F32Stream: fp32 I/O whose output stores are NON-TEMPORAL (`nt`) instead of write-through (`sc1`) -- for launches whose working set
// exceeds the 256 MB Infinity Cache.  Same-box A/B of the headline kernel (profiles/r05_ab_store_modifiers.txt): at 32768 x 64 (403 MB
// per launch) `nt` 66 us against `sc1` 84 us (0.76 against 0.60 of the HBM peak); inside the cache it is the other way round (4096 x 64:
// 11.4 against 9.15 us; 16384 x 64, 201 MB: 34.4 against 32.5), so the LAUNCH chooses (spec_stream_stores).  The HBM-side element type
// is a 4-byte wrapper, so that IoQuad -- where every store instruction of the generated kernels lives -- can be specialised on it.
struct trk_f32s {
    float v;
    __host__ __device__ trk_f32s() = default;
    __host__ __device__ trk_f32s(float x) : v(x) {}
    __host__ __device__ operator float() const { return v; }
};
struct F32Stream {};
template <> struct IoTraits<F32Stream> { typedef trk_f32s Q; typedef trk_f32s G; static constexpr bool kScaled = false; };
#define TRK_IO_F32 0
#define TRK_IO_F16 1
#define TRK_IO_F16_G32 2
// fp32 -> fp16 that saturates at the largest finite half instead of rounding to inf (v_med3_f32 + v_cvt_f16_f32)
__device__ __forceinline__ _Float16 trk_sat_f16(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
#ifndef TRK_EXP_PIECE_MOD
#define TRK_EXP_PIECE_MOD "sc1"     // experiment: the cache policy of the ring's 8-byte pieces
#endif
template <class IO> struct IoQuad;
template <> struct IoQuad<float> {
    static constexpr uintptr_t kAlignMask = 15;
    static __device__ __forceinline__ float4 load(const float* p, int k) { return reinterpret_cast<const float4*>(p)[k]; }
    static __device__ __forceinline__ void store_wt(float* p, int k, const float4& v) { store_wt_f4(reinterpret_cast<float4*>(p) + k, v); }
    // SGPR base + 32-bit per-lane byte offset: no 64-bit address arithmetic per lane
    static __device__ __forceinline__ void store_wt_s(unsigned long long base, unsigned voff, const float4& v) {
        const trk_f4 x = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(x), "s"(base)  TRK_STORE_CLOBBER);
    }
    // the same under a wave-uniform lane mask applied INSIDE the asm block (exec &= mask; store; restore): no control flow for
    // the compiler, so the store's LDS read is scheduled like any other load instead of sitting in a three-instruction branch
    // body right in front of its s_waitcnt
    static __device__ __forceinline__ void store_wt_sm(unsigned long long base, unsigned voff, const float4& v, unsigned long long mask) {
        const trk_f4 x = {v.x, v.y, v.z, v.w};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx4 %1, %2, %3 sc1\n s_mov_b64 exec, %0\n s_nop 0"
                     : "=&s"(saved) : "v"(voff), "v"(x), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1(float* p, float v) { store_wt_f1(p, v); }
    static __device__ __forceinline__ void store_wt_sat(float* p, int k, const float4& v) { store_wt(p, k, v); }
    static __device__ __forceinline__ void store_wt1_sat(float* p, float v) { store_wt_f1(p, v); }
    static __device__ __forceinline__ void store_wt2(float* p, float a, float b) { store_wt_f2(p, a, b); }
    static __device__ __forceinline__ void store_wt2_s(unsigned long long base, unsigned voff, float a, float b) {
        const trk_f2 x = {a, b};
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(x), "s"(base)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt1_s(unsigned long long base, unsigned voff, float a) {
        asm volatile("global_store_dword %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(a), "s"(base)  TRK_STORE_CLOBBER);
    }
    // the same under a lane mask applied INSIDE the asm block (exec &= mask; store; restore): no control flow for the compiler,
    // so a masked store does not end a scheduling region; an all-zero mask makes it a no-op
    static __device__ __forceinline__ void store_wt2_sm(unsigned long long base, unsigned voff, float a, float b, unsigned long long mask) {
        const trk_f2 x = {a, b};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx2 %1, %2, %3 " TRK_EXP_PIECE_MOD "\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(x), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1_sm(unsigned long long base, unsigned voff, float a, unsigned long long mask) {
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %3\n global_store_dword %1, %2, %4 sc1\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(a), "s"(mask), "s"(base) : "scc");
    }
};
template <> struct IoQuad<trk_f32s> {      // IoQuad<float> with `nt` stores (generated from it: keep the two in step)
    static constexpr uintptr_t kAlignMask = 15;
    static __device__ __forceinline__ float4 load(const trk_f32s* p, int k) { return reinterpret_cast<const float4*>(p)[k]; }
    static __device__ __forceinline__ void store_wt(trk_f32s* p, int k, const float4& v) { { const trk_f4 x = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off nt\n s_nop 1" :: "v"(reinterpret_cast<float4*>(p) + k), "v"(x)  TRK_STORE_CLOBBER); } }
    // SGPR base + 32-bit per-lane byte offset: no 64-bit address arithmetic per lane
    static __device__ __forceinline__ void store_wt_s(unsigned long long base, unsigned voff, const float4& v) {
        const trk_f4 x = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n s_nop 1" :: "v"(voff), "v"(x), "s"(base)  TRK_STORE_CLOBBER);
    }
    // the same under a wave-uniform lane mask applied INSIDE the asm block (exec &= mask; store; restore): no control flow for
    // the compiler, so the store's LDS read is scheduled like any other load instead of sitting in a three-instruction branch
    // body right in front of its s_waitcnt
    static __device__ __forceinline__ void store_wt_sm(unsigned long long base, unsigned voff, const float4& v, unsigned long long mask) {
        const trk_f4 x = {v.x, v.y, v.z, v.w};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx4 %1, %2, %3 nt\n s_mov_b64 exec, %0\n s_nop 0"
                     : "=&s"(saved) : "v"(voff), "v"(x), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1(trk_f32s* p, float v) { asm volatile("global_store_dword %0, %1, off nt\n s_nop 1" :: "v"(p), "v"(v)  TRK_STORE_CLOBBER); }
    static __device__ __forceinline__ void store_wt_sat(trk_f32s* p, int k, const float4& v) { store_wt(p, k, v); }
    static __device__ __forceinline__ void store_wt1_sat(trk_f32s* p, float v) { store_wt1(p, v); }
    static __device__ __forceinline__ void store_wt2(trk_f32s* p, float a, float b) { const trk_f2 x = {a, b}; asm volatile("global_store_dwordx2 %0, %1, off nt\n s_nop 1" :: "v"(p), "v"(x)  TRK_STORE_CLOBBER); }
    static __device__ __forceinline__ void store_wt2_s(unsigned long long base, unsigned voff, float a, float b) {
        const trk_f2 x = {a, b};
        asm volatile("global_store_dwordx2 %0, %1, %2 nt\n s_nop 1" :: "v"(voff), "v"(x), "s"(base)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt1_s(unsigned long long base, unsigned voff, float a) {
        asm volatile("global_store_dword %0, %1, %2 nt\n s_nop 1" :: "v"(voff), "v"(a), "s"(base)  TRK_STORE_CLOBBER);
    }
    // the same under a lane mask applied INSIDE the asm block (exec &= mask; store; restore): no control flow for the compiler,
    // so a masked store does not end a scheduling region; an all-zero mask makes it a no-op
    static __device__ __forceinline__ void store_wt2_sm(unsigned long long base, unsigned voff, float a, float b, unsigned long long mask) {
        const trk_f2 x = {a, b};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx2 %1, %2, %3 nt\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(x), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1_sm(unsigned long long base, unsigned voff, float a, unsigned long long mask) {
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %3\n global_store_dword %1, %2, %4 nt\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(a), "s"(mask), "s"(base) : "scc");
    }
};
template <> struct IoQuad<_Float16> {
    static constexpr uintptr_t kAlignMask = 7;
    static __device__ __forceinline__ float4 load(const _Float16* p, int k) {
        const trk_h4 h = reinterpret_cast<const trk_h4*>(p)[k];
        return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
    }
    static __device__ __forceinline__ void store_wt(_Float16* p, int k, const float4& v) {
        const trk_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(reinterpret_cast<trk_h4*>(p) + k), "v"(h)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt_s(unsigned long long base, unsigned voff, const float4& v) {
        const trk_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(h), "s"(base)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt_sm(unsigned long long base, unsigned voff, const float4& v, unsigned long long mask) {
        const trk_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx2 %1, %2, %3 sc1\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(h), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1(_Float16* p, float v) {
        const _Float16 h = (_Float16)v;
        asm volatile("global_store_short %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(h)  TRK_STORE_CLOBBER);
    }
    // gradient stores: saturating (the positions of a robot never leave the fp16 range, a scaled gradient may)
    static __device__ __forceinline__ void store_wt_sat(_Float16* p, int k, const float4& v) {
        const trk_h4 h = {trk_sat_f16(v.x), trk_sat_f16(v.y), trk_sat_f16(v.z), trk_sat_f16(v.w)};
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(reinterpret_cast<trk_h4*>(p) + k), "v"(h)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt1_sat(_Float16* p, float v) {
        const _Float16 h = trk_sat_f16(v);
        asm volatile("global_store_short %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(h)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt2(_Float16* p, float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = {(_Float16)a, (_Float16)b};
        asm volatile("global_store_dword %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(h)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt2_s(unsigned long long base, unsigned voff, float a, float b) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = {(_Float16)a, (_Float16)b};
        asm volatile("global_store_dword %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(h), "s"(base)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt1_s(unsigned long long base, unsigned voff, float a) {
        const _Float16 h = (_Float16)a;
        asm volatile("global_store_short %0, %1, %2 sc1\n s_nop 1" :: "v"(voff), "v"(h), "s"(base)  TRK_STORE_CLOBBER);
    }
    static __device__ __forceinline__ void store_wt2_sm(unsigned long long base, unsigned voff, float a, float b, unsigned long long mask) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 h = {(_Float16)a, (_Float16)b};
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dword %1, %2, %3 sc1\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(h), "s"(base), "s"(mask) : "scc");
    }
    static __device__ __forceinline__ void store_wt1_sm(unsigned long long base, unsigned voff, float a, unsigned long long mask) {
        const _Float16 h = (_Float16)a;
        unsigned long long saved;
        asm volatile("s_and_saveexec_b64 %0, %3\n global_store_short %1, %2, %4 sc1\n s_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(voff), "v"(h), "s"(mask), "s"(base) : "scc");
    }
};

// copy the first TRK_LDS_SPHERES world-frame spheres into this wave's LDS (one 16-byte load per lane).  Two halves: the load is
// issued before the wave's q rows and written to LDS after them, so its latency is the q loads' latency (as one function the
// write -- and with it an s_waitcnt vmcnt(0) -- sat in the branch body of the load, in front of the first q load).
struct SpheresInFlight { float4 v; bool on; };
__device__ __forceinline__ SpheresInFlight spec_load_spheres_issue(const DevCostHdr& C, int lane) {
    SpheresInFlight s;
    s.on = lane < TRK_LDS_SPHERES && lane < 2 * C.n_sphere_pairs;     // incl. the pad copy
    s.v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (s.on) s.v = C.spheres[lane];
    return s;
}
__device__ __forceinline__ void spec_load_spheres_finish(float4* lds_spheres, int lane, const SpheresInFlight& s) {
    if (s.on) lds_spheres[lane] = s.v;
}
// the same for the primitive table of a box scene (two float4s per record, lanes 0 .. 2 n_prims - 1; a table longer than
// TRK_LDS_PRIMS is not copied and scene_min_sdf keeps its select chain): lds_prims = the wave's TRK_LDS_PRIMS * 2 float4s
__device__ __forceinline__ SpheresInFlight spec_load_prims_issue(const DevCostHdr& C, int lane) {
    SpheresInFlight s;
    s.on = C.n_box_objects > 0 && C.n_prims <= TRK_LDS_PRIMS && lane < 2 * C.n_prims;
    s.v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (s.on) s.v = reinterpret_cast<const float4*>(C.prims)[lane];
    return s;
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

// q of sample (t, i, a) of the via-point mode: x[t, i, :D] * alpha[a] + x[t, i + 1, :D] * beta[a], each product and the sum rounded
// once like the reference's `a * alpha + b * (1 - alpha)` (trajectory/utils.py:47-49).  Neighbouring lanes read the same or
// adjacent way points, so the 2 D dword loads per lane are served by a few cache lines per wavefront.
template <int D>
__device__ __forceinline__ void spec_load_q_via(const SpecArgs& A, int64_t base, int rows, int lane, float (&qv)[D], unsigned& slot, int64_t& traj0,
                                                bool& outside) {
    const unsigned hi = (unsigned)(A.via_H - 1) * (unsigned)A.via_n;          // interpolated configurations per trajectory
    const int64_t t0 = base / hi;                                             // wave-uniform: one 64-bit division per wavefront
    const unsigned r = (unsigned)(base - t0 * hi) + (unsigned)lane;
    const unsigned dt = r / hi, rr = r - dt * hi;
    const unsigned i = rr / (unsigned)A.via_n, a = rr - i * (unsigned)A.via_n;
    const bool on = lane < rows;
    const float* p0 = static_cast<const float*>(A.q) + ((t0 + dt) * A.via_H + i) * (int64_t)A.via_S;
    const float fa = on ? A.via_alpha[a] : 0.0f, fb = on ? A.via_beta[a] : 0.0f;
    float pa[D], pb[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { pa[j] = on ? p0[j] : 0.0f; pb[j] = on ? p0[A.via_S + j] : 0.0f; }
#pragma unroll
    for (int j = 0; j < D; ++j) qv[j] = on ? __fadd_rn(__fmul_rn(pa[j], fa), __fmul_rn(pb[j], fb)) : 0.0f;
    slot = dt;                          // this lane's trajectory, counted from the wavefront's first one
    traj0 = t0;
    outside = false;
    if (A.via_partial) {                // wave-uniform: the joint-limit test of tasks.py:270-273 on way points this lane has loaded anyway.  Every
                                        // way point but a trajectory's last is the START of a segment: all lanes test theirs; the last one is
                                        // the END of the last segment -- tested only by the wavefronts that hold such a segment (wave-uniform
                                        // branch: one wavefront in ~60).  "v outside [lo, hi] or NaN" == "the BITS of clamp(v) differ from v's"
                                        // (v_med3 returns a bound for a NaN input; the generated units are compiled with -fno-honor-nans, so
                                        // no floating-point comparison is asked about a NaN): three instructions per value
        unsigned acc = 0u;
#pragma unroll
        for (int j = 0; j < D; ++j)
            acc |= __float_as_uint(__builtin_amdgcn_fmed3f(pa[j], cptr(A.via_qmin)[j], cptr(A.via_qmax)[j])) ^ __float_as_uint(pa[j]);
        const bool last_seg = on && i == (unsigned)(A.via_H - 2);
        if (__builtin_amdgcn_ballot_w64(last_seg) != 0ull) {
            unsigned accb = 0u;
#pragma unroll
            for (int j = 0; j < D; ++j)
                accb |= __float_as_uint(__builtin_amdgcn_fmed3f(pb[j], cptr(A.via_qmin)[j], cptr(A.via_qmax)[j])) ^ __float_as_uint(pb[j]);
            acc |= last_seg ? accb : 0u;
        }
        outside = acc != 0u && on;
    }
}
// The wavefront's share of the per-trajectory flags (see SpecArgs::via_partial): the byte of slot s -- trajectory t = traj0 + s, traj0 = the
// trajectory of the wavefront's first sample -- is written by lane s, and only when the wavefront holds samples of that trajectory.  The
// loop is wave-uniform (two trips when a trajectory has >= 64 interpolated configurations).
__device__ __forceinline__ void spec_via_partial_flags(const SpecArgs& A, int64_t wblock, int64_t traj0, unsigned slot, bool hit, bool outside,
                                                       bool on, int lane) {
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(on && hit), om = __builtin_amdgcn_ballot_w64(on && outside);
    unsigned mine = 0u;
    bool have = false;
    for (int s = 0; s < A.via_slots; ++s) {
        const unsigned long long grp = __builtin_amdgcn_ballot_w64(on && slot == (unsigned)s);
        const unsigned bits = ((hm & grp) ? 1u : 0u) | ((om & grp) ? 2u : 0u);
        mine = lane == s ? bits : mine;
        have = lane == s ? grp != 0ull : have;
    }
    if (have) {
        const int64_t hi = (int64_t)(A.via_H - 1) * A.via_n;
        const int64_t t = traj0 + lane;                                      // lane == slot here
        A.via_partial[t * (hi / TRK_WAVE + 2) + (wblock - ((t * hi) >> 6))] = (uint8_t)mine;
    }
}

// spec_load_q in two halves: `issue` starts the wave's 16-byte loads (held in registers, nothing waits), `finish` runs the LDS
// transpose.  Whatever the kernel computes in between overlaps the HBM latency (k_posbwd: the forward pass needs only q, the
// adjoint rows are not touched before the reverse pass).
template <int D>
struct RowsInFlight {
    static constexpr int NV = TRK_WAVE * D / 4, NJ = (NV + TRK_WAVE - 1) / TRK_WAVE;
    float4 v[NJ];
    bool fast;
};
template <int D, class IO>
__device__ __forceinline__ RowsInFlight<D> spec_load_rows_issue(const IO* __restrict__ in, int64_t base, int rows, int lane) {
    RowsInFlight<D> r;
    const IO* src = in + base * D;
    r.fast = rows == TRK_WAVE && (TRK_WAVE * D) % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) & IoQuad<IO>::kAlignMask) == 0);
#pragma unroll
    for (int j = 0; j < RowsInFlight<D>::NJ; ++j) {
        const int k = lane + TRK_WAVE * j;
        r.v[j] = (r.fast && k < RowsInFlight<D>::NV) ? IoQuad<IO>::load(src, k) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    return r;
}
template <int D, class IO>
__device__ __forceinline__ void spec_load_rows_finish(const RowsInFlight<D>& r, const IO* __restrict__ in, int64_t base, int rows,
                                                      int lane, float* lds, float (&qv)[D]) {
    if (!r.fast) { spec_load_q<D>(in, base, rows, lane, lds, qv); return; }      // ragged / unaligned: the one-step path, now
    spec_wave_sync();
    float4* lds4 = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int j = 0; j < RowsInFlight<D>::NJ; ++j) {
        const int k = lane + TRK_WAVE * j;
        if (k < RowsInFlight<D>::NV) lds4[k] = r.v[j];
    }
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) qv[j] = lds[lane * D + j];
    spec_wave_sync();
}

// SCALED: the values are multiplied by `scale` on their way into the tile and an fp16 store saturates (IoTraits<IO>::kScaled)
template <int D, class IO, bool SCALED = false>
__device__ __forceinline__ void spec_store_gq(IO* __restrict__ gq, int64_t base, int rows, int lane,
                                              float* lds, const float (&gv)[D], float scale = 1.0f) {
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < D; ++j) lds[lane * D + j] = SCALED ? gv[j] * scale : gv[j];
    spec_wave_sync();
    const int64_t first = base * D;
    IO* dst = gq + first;
    constexpr int NV = TRK_WAVE * D / 4;
    if (rows == TRK_WAVE && (TRK_WAVE * D) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0)) {
        const float4* lds4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
        for (int j = 0; j < (NV + TRK_WAVE - 1) / TRK_WAVE; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < NV) { if (SCALED) IoQuad<IO>::store_wt_sat(dst, k, lds4[k]); else IoQuad<IO>::store_wt(dst, k, lds4[k]); }
        }
    } else {
        const int count = rows * D;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < count) { if (SCALED) IoQuad<IO>::store_wt1_sat(dst + k, lds[k]); else IoQuad<IO>::store_wt1(dst + k, lds[k]); }
        }
    }
}

// copy this wave's [64][W] tile (rows contiguous in LDS, 16-byte aligned) to out[base .. base + rows) as 16-byte write-through stores
// (NT: non-temporal instead -- for outputs of launches whose working set exceeds the Infinity Cache)
template <int W, bool NT = false>
__device__ __forceinline__ void spec_store_tile(float* __restrict__ out, int64_t base, int rows, int lane, const float* tile) {
    float* dst = out + base * W;
    constexpr int NV = TRK_WAVE * W / 4;
    if (rows == TRK_WAVE && (TRK_WAVE * W) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
#pragma unroll
        for (int j = 0; j < (NV + TRK_WAVE - 1) / TRK_WAVE; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < NV) { if (NT) IoQuad<trk_f32s>::store_wt(reinterpret_cast<trk_f32s*>(dst), k, t4[k]); else IoQuad<float>::store_wt(dst, k, t4[k]); }
        }
    } else {
        const int count = rows * W;
        for (int k = lane; k < count; k += TRK_WAVE) dst[k] = tile[k];
    }
}

// Link positions: each lane writes its 3L floats at stride 3L (33 for Panda: conflict-free), then the wave
// streams the 64*3L contiguous floats out in 1 KiB chunks (one ds_read_b128 + one global_store_dwordx4 per
// lane and chunk).  The chunks are NOT issued back to back: every wave of the chip reaches this point at the same
// time, and 16 waves x 8.25 KiB per CU saturate the store path for ~6 us during which nothing computes.
// `tick()` issues one chunk; the kernel calls it between blocks of arithmetic so the 34.6 MB trickle out at
// roughly the rate HBM absorbs them.
// ---- analytic Jacobian of every link (generated k_ajac; the table-driven kernel is k_fk_analytic_jacobian): a link's 7 x D block, rows
// 0 .. 2 = d pos / d q, rows 3 .. 6 = d quat_wxyz / d q; a sample's row of the output is its L blocks back to back (49 L floats for 7 DOF),
// staged through the ring (RingFlusher) in memory order.
// Column d of a link whose chain holds the REVOLUTE joint d: omega = pass * sign * (the joint's axis in the world), pj = the joint link's
// origin; d p = omega x (p - pj), d R = [omega]x R -> d quat through the selected candidate of rotation_matrix_to_q (quat_jvp).
__device__ __forceinline__ void spec_ajac_col_revolute(float (&c)[7], const float (&Ri)[9], const QuatSel& qs, float t0, float t1, float t2,
                                                       float w0, float w1, float w2, float p0, float p1, float p2) {
    const float r0 = t0 - p0, r1 = t1 - p1, r2 = t2 - p2;
    c[0] = w1 * r2 - w2 * r1; c[1] = w2 * r0 - w0 * r2; c[2] = w0 * r1 - w1 * r0;
    float dR[9], dq[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v0 = Ri[k], v1 = Ri[3 + k], v2 = Ri[6 + k];
        dR[k] = w1 * v2 - w2 * v1; dR[3 + k] = w2 * v0 - w0 * v2; dR[6 + k] = w0 * v1 - w1 * v0;
    }
    quat_jvp_sel(qs, dR, dq);       // the candidate of rotation_matrix_to_q was selected once for the link (quat_sel)
    c[3] = dq[0]; c[4] = dq[1]; c[5] = dq[2]; c[6] = dq[3];
}

template <int W, class IO>
struct PosFlusher {
    static constexpr int NV = W * TRK_WAVE / 4;                  // 4-element vectors per wave
    static constexpr int NCHUNK = (NV + TRK_WAVE - 1) / TRK_WAVE;
    static constexpr int TAIL = NV - (NCHUNK - 1) * TRK_WAVE;    // lanes of the last chunk (64 = full)
    static constexpr int CB = TRK_WAVE * 4 * sizeof(IO);         // bytes of one chunk in HBM
    // Chunks are numbered at COMPILE time (the generator knows where every tick sits), so a tick is: two scalar adds for the
    // chunk's address, one ds_read_b128 at an immediate offset, one store with an SGPR base -- no VALU work at all.  (The
    // first version carried a chunk counter and rebuilt a 64-bit per-lane address and an exec mask per tick: ~9 VALU
    // instructions, most of them in the 4-cycle class, 9 ticks per wavefront.)
    const float4* src;                                           // this lane's vector of chunk 0, in LDS (always a readable address)
    unsigned voff;                                               // lane * bytes per vector in HBM
    unsigned long long g0;                                       // wave-uniform: address of the wave's first byte
    unsigned long long mask;                                     // wave-uniform lane mask of the stores: every lane, or none (nothing staged / already written)
    int lane;
    mutable float4 nxt;                                          // the next chunk's vector, read from LDS one tick ahead
    // Straight-line: the lane mask is applied inside the store's asm block, and the LDS read of chunk CH + 1 is issued BEFORE the
    // store of chunk CH (the stores are `asm volatile`, which no memory operation is scheduled across: read where it is used, a
    // chunk's ds_read_b128 sat right in front of its own s_waitcnt + store, ~130 cycles of LDS latency exposed at every tick).
    // Chunks are taken in order, each once (every path of the kernels does).
    __device__ __forceinline__ const float4* vec(int ch) const {
        // lanes past the tile in the partial last chunk read their chunk-0 vector instead (their store is masked off)
        return (ch < NCHUNK - 1 || TAIL == TRK_WAVE || lane < TAIL) ? src + ch * TRK_WAVE : src;
    }
    __device__ __forceinline__ void prime() const { nxt = *vec(0); }
    template <int CH>
    __device__ __forceinline__ void chunk() const {
        if constexpr (CH >= 0 && CH < NCHUNK) {
            const unsigned long long g = g0 + (unsigned long long)CH * CB;
            const float4 v = nxt;
            if constexpr (CH + 1 < NCHUNK) nxt = *vec(CH + 1);
            IoQuad<IO>::store_wt_sm(g, voff, v, (CH < NCHUNK - 1 || TAIL == TRK_WAVE) ? mask : (mask & ((1ull << (TAIL & 63)) - 1ull)));
        }
    }
    template <int A, int B>                                      // chunks A .. B-1
    __device__ __forceinline__ void range() const {
        if constexpr (A < B && A < NCHUNK) { chunk<A>(); range<A + 1, B>(); }
    }
    template <int A> __device__ __forceinline__ void rest() const { range<A, NCHUNK>(); }
};

// What the scene evaluation is handed: its tick slots `at<0>() .. at<TRK_OBJ_TICK_SLOTS - 1>()` = chunks BASE .. of the flusher.
template <class F, int BASE>
struct TickFrom {
    const F& f;
    template <int J> __device__ __forceinline__ void at() const { f.template chunk<BASE + J>(); }
};

// stand-in for PosFlusher in kernels whose positions leave through spec_flush_chunk instead
struct NoFlush {
    template <int CH> __device__ __forceinline__ void chunk() const {}
    template <int A, int B> __device__ __forceinline__ void range() const {}
    template <int A> __device__ __forceinline__ void rest() const {}
};

template <class R> struct NoFlushOf : NoFlush {      // the same, constructible from a flusher it ignores
    __device__ __forceinline__ NoFlushOf(const R&) {}
};

template <int W, class IO>
__device__ __forceinline__ PosFlusher<W, IO> spec_make_flusher(IO* __restrict__ out, int64_t base, int rows, int lane, float* lds) {
    IO* dst = out + base * W;
    const bool fast = rows == TRK_WAVE && (W * TRK_WAVE) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0);
    if (!fast) {                                                 // ragged last wavefront / unaligned view: plain copy, now
        const int count = rows * W;
        for (int k = lane; k < count; k += TRK_WAVE) dst[k] = (IO)lds[k];
    }
    // the address is wave-uniform by construction; say so, so that it is held in SGPRs
    const unsigned long long g = (unsigned long long)reinterpret_cast<uintptr_t>(dst);
    const unsigned long long gu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)g);
    // a ballot is scalar by definition (an SGPR pair): all lanes, or none
    PosFlusher<W, IO> f{reinterpret_cast<const float4*>(lds) + lane, (unsigned)(lane * 4 * sizeof(IO)), gu, __builtin_amdgcn_ballot_w64(fast), lane,
                        make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    f.prime();
    return f;
}

// stage the wave's rows in LDS; returns a flusher (fast path) or writes everything now (ragged / unaligned tail)
template <int W, class IO>
__device__ __forceinline__ PosFlusher<W, IO> spec_stage_rows(IO* __restrict__ out, int64_t base, int rows, int lane,
                                                             float* lds, const float (&v)[W]) {
    spec_wave_sync();
#pragma unroll
    for (int j = 0; j < W; ++j) lds[lane * W + j] = v[j];
    spec_wave_sync();
    return spec_make_flusher<W, IO>(out, base, rows, lane, lds);
}

// a wave-uniform address, said to be so: held in an SGPR pair
template <class T>
__device__ __forceinline__ unsigned long long spec_uniform_address(T* p) {
    const unsigned long long g = (unsigned long long)reinterpret_cast<uintptr_t>(p);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)g);
}

// lane -> (sample, vector) bookkeeping of the chunk copies below: vector j of a lane is element lane + 64 j of a [64][NVEC] block
template <int NVEC>
struct ChunkLane {
    int s0, r0;                 // lane = NVEC s0 + r0
    __device__ __forceinline__ explicit ChunkLane(int lane) : s0(lane / NVEC), r0(lane - (lane / NVEC) * NVEC) {}
    // index of vector j when samples are S elements apart and a vector is V elements: (s0 + a + c) S + (r0 + b - c NVEC) V, given
    // l0 = s0 S + r0 V; j is a constant after unrolling, so a, b and both products are immediates
    template <int S, int V>
    __device__ __forceinline__ int offset(int l0, int j) const {
        const int a = (TRK_WAVE * j) / NVEC, b = (TRK_WAVE * j) % NVEC;
        return l0 + (a * S + b * V) + ((b != 0 && r0 >= NVEC - b) ? S - NVEC * V : 0);
    }
    template <int S, int V, int BYTES>
    __device__ __forceinline__ unsigned offset_bytes(unsigned g0, int j) const {
        const int a = (TRK_WAVE * j) / NVEC, b = (TRK_WAVE * j) % NVEC;
        return g0 + (unsigned)((a * S + b * V) * BYTES) + ((b != 0 && r0 >= NVEC - b) ? (unsigned)((S - NVEC * V) * BYTES) : 0u);
    }
};

// Wide rows (attached points: W = 3P floats per sample) do not fit a whole-row staging buffer, so they leave in column
// chunks: a chunk is the NF consecutive floats [c0, c0 + NF) of every sample's row.  Each lane has put its NF floats at
// lds[lane * LS ...]; the wave then streams the rows' segments as V-float vectors (V | NF, V | W, V | c0: 16/8/4-byte
// aligned), vector e -> sample e / (NF / V).  A sample's segment is contiguous, neighbouring chunks complete its lines.
// UNALIGNED: the row length is not a multiple of the vector (the generator's choice for fp32 rows: a wide store whose address is only 4- or
// 8-byte aligned is legal on this chip and halves / quarters the store instructions of such rows); the LDS side stays aligned.
template <int W, int NF, int LS, int V, class IO = float, bool UNALIGNED = false>
__device__ __forceinline__ void spec_flush_chunk(IO* __restrict__ out, int64_t base, int c0, int rows, int lane,
                                                 const float* lds) {
    constexpr int NVEC = NF / V;
    static_assert(NF % V == 0 && (UNALIGNED || W % V == 0) && (V != 4 || LS % 4 == 0), "chunk geometry must keep the vectors aligned");
    spec_wave_sync();
    const int total = rows * NVEC;
    IO* dst0 = out + base * W + c0;
    if (rows == TRK_WAVE) {      // full wave: a batch of LDS reads in flight before the first store (guarded, every read sits in a
        constexpr int NB = NVEC * V > 36 ? (NVEC + 3) / 4 : NVEC;     // branch body of its own right in front of its s_waitcnt)
        // Vector j of a lane is e = lane + 64 j -> (sample e / NVEC, vector e % NVEC).  As a division per vector that was ~14 integer
        // instructions per store (multiply-high, shifts, a 64-bit address: 546 of the grasped-box kernel's 2827 vector instructions per
        // wavefront, 2980 of 7231 with the 213 four-byte stores of the 71-point model).  Incrementally: 64 j = NVEC a + b at compile
        // time, so sample = s0 + a + c and vector = r0 + b - c NVEC with c = (r0 >= NVEC - b) -- one compare and a select per offset,
        // the wave-uniform destination in SGPRs (store with a 32-bit lane offset).
        const ChunkLane<NVEC> cl(lane);
        const int l0 = cl.s0 * LS + cl.r0 * V;
        const unsigned g0 = (unsigned)((cl.s0 * W + cl.r0 * V) * (int)sizeof(IO));
        const unsigned long long gu = spec_uniform_address(dst0);
#pragma unroll
        for (int j0 = 0; j0 < NVEC; j0 += NB) {
            float a[NB][V];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j0 + j < NVEC) {
                    const float* src = lds + cl.template offset<LS, V>(l0, j0 + j);
                    if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(src); a[j][0] = t.x; a[j][1] = t.y; a[j][2 % V] = t.z; a[j][3 % V] = t.w; }
                    else {
#pragma unroll
                        for (int k = 0; k < V; ++k) a[j][k] = src[k];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j0 + j < NVEC) {
                    const unsigned voff = cl.template offset_bytes<W, V, (int)sizeof(IO)>(g0, j0 + j);
                    if (V == 4) IoQuad<IO>::store_wt_s(gu, voff, make_float4(a[j][0], a[j][1], a[j][2 % V], a[j][3 % V]));
                    else if (V == 2) IoQuad<IO>::store_wt2_s(gu, voff, a[j][0], a[j][1 % V]);
                    else IoQuad<IO>::store_wt1_s(gu, voff, a[j][0]);
                }
            }
        }
        spec_wave_sync();
        return;
    }
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

// Ring staging for link kernels with many links.  spec_flush_chunk issues a chunk's stores back to back: every wave of the chip
// reaches that point together, the store path fills, and nothing computes until the burst has drained (UR10+Allegro, 4096 x 64:
// 30.0 us with positions against 13.5 us without -- exactly the 94 MB at HBM write speed, no overlap).  Here every lane stages
// its sample's floats in a 64-float RING in LDS (two 32-float chunks: one being filled, one leaving) and a complete chunk
// leaves as single store instructions ("pieces") that the generator spreads over the code that follows -- the links that fill
// the other half of the ring, then the objectives.
//
// Sector alignment (ALIGNED): a row of W floats starts at byte 4 W s, in general in the middle of a 32-byte HBM sector, so
// fixed column chunks [32c, 32c + 32) split two sectors per sample and chunk between different store instructions (measured:
// WRITE_SIZE 137 MB for 118 MB of output).  Instead sample s owns a HEAD of h(s) = (-W s) mod 8 floats -- the part of its row
// that shares a sector with the end of row s - 1 -- and its chunk c is the floats [h + 32c, h + 32c + 32): four whole sectors.
// Ring position of float f: (f - h) & 63.  What is left of a row is its TAIL, T - h floats; the tail of row s and the head of
// row s + 1 are contiguous in memory and together a whole number of sectors (T - g or T - g + 8 floats, g = W mod 8), and they
// leave together as one UNIT: every sector of the output is written exactly once, by one instruction.  The head floats are
// kept twice (ring + HX extra floats per lane) because their ring slots are overwritten before the tail leaves; a unit reads the
// head from the NEXT lane's row.
//
// Piece K of a chunk covers the S = 64 / (32 / V) samples [K S, K S + S), lane -> (sample lane / NVEC, vector lane % NVEC): a
// piece costs a few integer instructions, an LDS read at an immediate offset and one store with an SGPR base.
// Unaligned rings (ALIGNED = false: h = 0 everywhere) serve the row lengths whose tail unit would exceed a chunk.
template <int W, int V, bool ALIGNED, class IO>
struct RingFlusher {
    static_assert(V == 1 || V == 2, "ring pieces are 1- or 2-float vectors (the ring stride is odd)");
    static_assert(!ALIGNED || V == 2, "sector-aligned chunks leave as 2-float vectors");
    static constexpr int CS = 32;                                   // floats per chunk
    static constexpr int G = W % 8;                                 // a row start advances by G floats within its sector
    static constexpr int HX = !ALIGNED || G == 0 ? 0 : (G % 2 ? 7 : (G == 4 ? 4 : 6));   // longest head
    static constexpr int LS = (64 + HX) | 1;                        // per-lane stride: odd -> the per-lane row writes are conflict-free
    static constexpr int NFULL = (W - (HX > 1 ? HX : 1)) / CS;      // whole chunks of every sample (the tail is never empty)
    static constexpr int T = W - CS * NFULL;                        // floats of a sample's tail + head
    static constexpr int NVEC = CS / V, S = TRK_WAVE / NVEC, NP = TRK_WAVE / S;   // vectors per sample, samples per piece, pieces per chunk
    static constexpr int UMAX = ALIGNED && G ? T - G + 8 : T;       // longest tail unit
    static constexpr int NVT = (UMAX + V - 1) / V;
    static_assert(UMAX % V == 0 && T >= HX && UMAX <= CS && W >= CS, "row length not supported by the ring geometry");
    IO* out;                    // nullptr: positions not wanted
    int64_t base;
    int rows, lane;
    float* lds;                 // this wave's ring: [64][LS]
    unsigned long long g0;      // wave-uniform: address of the wave's first output element
    bool fast;                  // wave-uniform: full wave and aligned rows -> pieces; else every chunk is copied when complete
    unsigned long long pieces_on;   // wave-uniform lane mask of the pieces: all lanes, or none (no output / copy path)
    static __device__ __forceinline__ int head(int smp) { return ALIGNED ? ((-W * smp) & 7) : 0; }
    // this lane's staging pointers: regular floats go to row_a()[f & 63], the few whose ring slot depends on the head
    // (f < HX or (f & 63) < HX) to row()[(f - head) & 63], floats f < HX additionally to row()[64 + f]
    __device__ __forceinline__ float* row() const { return lds + lane * LS; }
    __device__ __forceinline__ float* row_a() const { return lds + lane * LS - head(lane); }
    __device__ __forceinline__ int slot(int f) const { return (f - head(lane)) & 63; }

    // chunk C (C == NFULL: the tail) is complete in the ring
    template <int C>
    __device__ __forceinline__ void done() const {
        spec_wave_sync();
        if (out && !fast) {                                          // ragged last wavefront / unaligned view: plain copy, now
            if constexpr (C < NFULL) {
                for (int e = lane; e < rows * CS; e += TRK_WAVE) {
                    const int smp = e / CS, j = e - smp * CS;
                    out[(base + smp) * W + head(smp) + CS * C + j] = (IO)lds[smp * LS + ((CS * C + j) & 63)];
                }
            } else {                                                 // every sample writes its own tail and its own head
                for (int e = lane; e < rows * T; e += TRK_WAVE) {
                    const int smp = e / T, j = e - smp * T, h = head(smp);
                    if (j < T - h) out[(base + smp) * W + h + CS * NFULL + j] = (IO)lds[smp * LS + ((CS * NFULL + j) & 63)];
                    else out[(base + smp) * W + (j - (T - h))] = (IO)lds[smp * LS + 64 + (j - (T - h))];
                }
            }
            spec_wave_sync();
        }
    }
    // Store piece K of chunk C.  Straight-line code: the lane mask (nothing when the positions are not wanted or the wave took
    // the copy path) is applied inside the store's asm block -- 48 wave-uniform branches would cut the FK arithmetic into 48
    // scheduling regions (measured on the launches WITHOUT positions: +2 us).  The piece's offset goes into the per-lane offset
    // (one VALU add): as a scalar add to the base the scheduler hoists 48 address pairs to the top and spills them.
    // Reading a piece one slot before storing it, so that the LDS latency hides behind a link's arithmetic, was measured: no
    // change (32.2 vs 31.9 us) -- the stores, not the reads, are what the wave waits for.
    template <int C, int K>
    __device__ __forceinline__ void piece() const {
        static_assert(C >= 0 && C <= NFULL, "no such chunk");
        if constexpr (K >= 0 && K < NP) {
            const float* r0 = lds + K * S * LS;                      // row of the piece's first sample
            if constexpr (C < NFULL) {
                const int ds = lane / NVEC, v = lane - ds * NVEC;
                const float* s = r0 + ds * LS + ((CS * C) & 63) + v * V;
                const int elem = (K * S + ds) * W + head(K * S + ds) + CS * C + v * V;
                put(elem, s[0], V == 2 ? s[1] : 0.0f, pieces_on);
            } else {
                // tail unit of sample s = K S + ds: its last T - h(s) floats, then the head of row s + 1 (contiguous in memory)
                const int dq = lane / NVT, v = lane - dq * NVT;
                const int ds = dq < S ? dq : S - 1;                  // lanes past the piece's S units are masked off; keep their reads in range
                const int h = head(K * S + ds), hn = (K * S + ds + 1 < TRK_WAVE) ? head(K * S + ds + 1) : 0;
                const int nt = T - h;                                // floats that come from this row's ring
                const int j = v * V;
                const float* own = r0 + ds * LS;
                const float* nxt = own + LS + 64;                    // the next lane's head copy
                const float a = j < nt ? own[(CS * NFULL + j) & 63] : nxt[j - nt];
                const float b = V == 2 ? (j + 1 < nt ? own[(CS * NFULL + j + 1) & 63] : nxt[j + 1 - nt]) : 0.0f;
                const bool on = dq < S && j < nt + hn;
                const int elem = (K * S + ds) * W + h + CS * NFULL + j;
                put(elem, a, b, pieces_on & __builtin_amdgcn_ballot_w64(on));
            }
        }
    }
    __device__ __forceinline__ void put(int elem, float a, float b, unsigned long long mask) const {
        const unsigned voff = (unsigned)(elem * (int)sizeof(IO));
        if (V == 2) IoQuad<IO>::store_wt2_sm(g0, voff, a, b, mask);
        else IoQuad<IO>::store_wt1_sm(g0, voff, a, mask);
    }
};

template <int W, int V, bool ALIGNED, class IO>
__device__ __forceinline__ RingFlusher<W, V, ALIGNED, IO> spec_make_ring(IO* __restrict__ out, int64_t base, int rows, int lane, float* lds) {
    IO* dst = out + base * W;
    const bool fast = rows == TRK_WAVE && ((reinterpret_cast<uintptr_t>(dst) & (V * sizeof(IO) - 1)) == 0);
    const unsigned long long g = (unsigned long long)reinterpret_cast<uintptr_t>(dst);
    const unsigned long long gu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)g);
    // a ballot is scalar by definition (an SGPR pair): all lanes, or none
    return RingFlusher<W, V, ALIGNED, IO>{out, base, rows, lane, lds, gu, fast, __builtin_amdgcn_ballot_w64(out != nullptr && fast)};
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
    if (rows == TRK_WAVE) {      // full wave: a batch of loads in flight before the first LDS write (the guarded
        constexpr int NB = NVEC * V > 36 ? (NVEC + 3) / 4 : NVEC;     // loop below waits for every load before it issues the next)
        typedef typename TrkIf<V == 4, float4, typename TrkIf<V == 2, float2, float>::type>::type vec_t;
        const ChunkLane<NVEC> cl(lane);        // (sample, vector) of a lane's vectors without a division per vector: see spec_flush_chunk
        const int l0 = cl.s0 * LS + cl.r0 * V, g0 = cl.s0 * W + cl.r0 * V;
#pragma unroll
        for (int j0 = 0; j0 < NVEC; j0 += NB) {
            vec_t r[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j0 + j < NVEC) r[j] = *reinterpret_cast<const vec_t*>(src0 + cl.template offset<W, V>(g0, j0 + j));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j0 + j < NVEC) *reinterpret_cast<vec_t*>(lds + cl.template offset<LS, V>(l0, j0 + j)) = r[j];
            }
        }
        spec_wave_sync();
        return;
    }
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
    return spec_make_flusher<W, IO>(out, base, rows, lane, lds);
}

// ---------------------------------------------------------------------------------------------------------
// Pieces of the GP-fused rollout (k_rollout_gp): the wave's rows of q / qd as a RAW tile (HBM element type kept: an fp16 trajectory
// costs 2 bytes per element of LDS), the rows of the two neighbouring samples next to it, and position blocks that leave segment
// by segment.
// ---------------------------------------------------------------------------------------------------------
// Layout of a raw tile: PAD elements (a multiple of 16 bytes, >= D), then the 64 rows of the block, then one more row:
//   body = tile + PAD (16-byte aligned);  body[r * D + j] = src[(base + r) * D + j] for r = -1 .. 64
// rows -1 and 64 are the previous / next sample of the trajectory when it continues beyond this wavefront's block (else zero; the
// caller masks them by the time step anyway).  Two halves like spec_load_rows_issue / finish: every load of both tiles is in
// flight before the first LDS write.
// ROWS (default 64): rows of the block.  The arm-per-lane kernels (k_rollout_gpa) give a wavefront 32 samples -- two lanes per sample --
// and read the same tile through the (2 ROWS, D / 2) view: lane l's half row starts at element l * (D / 2).
template <int D, class IO, int ROWS = TRK_WAVE>
struct RawRowsInFlight {
    static constexpr int VE = 16 / sizeof(IO);                          // elements per 16-byte vector
    static constexpr int PAD = (D + VE - 1) / VE * VE;
    static constexpr int ELEMS = PAD + (ROWS + 1) * D;                  // elements of a tile
    static constexpr int BYTES = (ELEMS * (int)sizeof(IO) + 15) / 16 * 16;
    static constexpr int NV = ROWS * D / VE, NJ = (NV + TRK_WAVE - 1) / TRK_WAVE;
    static constexpr int NSLOW = (ROWS * D + TRK_WAVE - 1) / TRK_WAVE;  // element-wise trips of the ragged / unaligned path
    trk_f4 v[NJ];
    IO edge;                                                            // lanes 0 .. D-1: row -1; lanes 32 .. 32+D-1: row ROWS
    bool fast;
};
template <int D, class IO, int ROWS = TRK_WAVE>
__device__ __forceinline__ RawRowsInFlight<D, IO, ROWS> spec_raw_rows_issue(const IO* __restrict__ in, int64_t base, int rows, int lane,
                                                                            bool has_prev, bool has_next) {
    static_assert(D <= 32, "edge rows are fetched by one half-wave each");
    typedef RawRowsInFlight<D, IO, ROWS> Raw;
    Raw r;
    const IO* src = in + base * D;
    r.fast = rows == ROWS && (ROWS * D * sizeof(IO)) % 16 == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
#pragma unroll
    for (int j = 0; j < Raw::NJ; ++j) {
        const int k = lane + TRK_WAVE * j;
        r.v[j] = (r.fast && k < Raw::NV) ? reinterpret_cast<const trk_f4*>(src)[k] : trk_f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    r.edge = (IO)0.0f;
    if (lane < D && has_prev) r.edge = src[lane - D];                   // the D elements in front of the block
    if (lane >= 32 && lane < 32 + D && has_next) r.edge = src[ROWS * D + (lane - 32)];
    return r;
}
template <int D, class IO, int ROWS = TRK_WAVE>
__device__ __forceinline__ IO* spec_raw_rows_finish(const RawRowsInFlight<D, IO, ROWS>& r, const IO* __restrict__ in, int64_t base, int rows,
                                                    int lane, IO* tile) {
    typedef RawRowsInFlight<D, IO, ROWS> Raw;
    IO* body = tile + Raw::PAD;
    if (r.fast) {
        trk_f4* b4 = reinterpret_cast<trk_f4*>(body);
#pragma unroll
        for (int j = 0; j < Raw::NJ; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < Raw::NV) b4[k] = r.v[j];
        }
    } else {
        const IO* src = in + base * D;
        const int count = rows * D;
#pragma unroll
        for (int j = 0; j < Raw::NSLOW; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < ROWS * D) body[k] = k < count ? src[k] : (IO)0.0f;
        }
    }
    if (lane < D) body[lane - D] = r.edge;
    if (lane >= 32 && lane < 32 + D) body[ROWS * D + (lane - 32)] = r.edge;
    return body;
}

// the wave's [64][D] fp32 tile (each lane wrote its own row) -> out rows [base, base + rows), multiplied by `scale` (SCALED)
template <int D, class IO, bool SCALED>
__device__ __forceinline__ void spec_store_acc_tile(IO* __restrict__ out, int64_t base, int rows, int lane, const float* tile, float scale) {
    spec_wave_sync();
    IO* dst = out + base * D;
    constexpr int NV = TRK_WAVE * D / 4;
    if (rows == TRK_WAVE && (TRK_WAVE * D) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & IoQuad<IO>::kAlignMask) == 0)) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
#pragma unroll
        for (int j = 0; j < (NV + TRK_WAVE - 1) / TRK_WAVE; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < NV) {
                float4 v = t4[k];
                if (SCALED) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; IoQuad<IO>::store_wt_sat(dst, k, v); }
                else IoQuad<IO>::store_wt(dst, k, v);
            }
        }
    } else {
        const int count = rows * D;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int k = lane + TRK_WAVE * j;
            if (k < count) { if (SCALED) IoQuad<IO>::store_wt1_sat(dst + k, tile[k] * scale); else IoQuad<IO>::store_wt1(dst + k, tile[k]); }
        }
    }
}

// The positions of the GP-fused kernel: every lane stages its sample's elements AS HBM ELEMENTS (IO) at their place in the
// wavefront's output image -- img[lane * W + c], the 64 rows back to back exactly as they lie in HBM (an fp16 image is half the
// size of the fp32 tile spec_stage_rows keeps: 8.8 KB for the dual Panda) -- segment by segment; once the last segment is staged
// the image leaves as contiguous 16-byte vectors, one LDS read at an immediate offset and one store with an SGPR base per piece.
// (First version: each segment's columns left on their own as 4-byte pieces -- a sample's block is contiguous but the blocks of
// one store instruction are not, every 64-byte line was written in parts by several instructions: 7.7 us for the dual Panda's
// 36 MB where these whole-line stores take 2.5; as write-back stores 26 us more.)
template <int W, class IO, int ROWS = TRK_WAVE>
struct ImgFlusher {
    static constexpr int BYTES = ROWS * W * (int)sizeof(IO);
    static constexpr int NV = BYTES / 16;                                // whole 16-byte vectors
    static constexpr int NP = (NV + TRK_WAVE - 1) / TRK_WAVE;            // store instructions
    static constexpr int TAIL = NV - (NP - 1) * TRK_WAVE;                // lanes of the last one
    static_assert(BYTES % 16 == 0, "the rows of a wavefront must be whole 16-byte vectors");
    IO* img;                                                             // this wave's image
    const trk_f4* src;                                                   // this lane's vector of piece 0
    unsigned voff;                                                       // lane * 16
    unsigned long long g0;                                               // wave-uniform: address of the wave's first output byte
    unsigned long long on;                                               // wave-uniform: all lanes (full wavefront, positions wanted) or none
    int lane;
    // a lane's own row, element c (one lane per row: ROWS == 64)
    __device__ __forceinline__ void put(int c, float v) const { img[lane * W + c] = (IO)v; }
    template <int J>
    __device__ __forceinline__ void piece() const {
        if constexpr (J >= 0 && J < NP) {
            const trk_f4 v = (J < NP - 1 || TAIL == TRK_WAVE || lane < TAIL) ? src[J * TRK_WAVE] : src[0];
            const unsigned long long g = g0 + (unsigned long long)J * (TRK_WAVE * 16);
            const unsigned long long m = (J < NP - 1 || TAIL == TRK_WAVE) ? on : (on & ((1ull << (TAIL & 63)) - 1ull));
            unsigned long long saved;
            asm volatile("s_and_saveexec_b64 %0, %4\n global_store_dwordx4 %1, %2, %3 sc1\n s_mov_b64 exec, %0\n s_nop 0"
                         : "=&s"(saved) : "v"(voff), "v"(v), "s"(g), "s"(m) : "scc");
        }
    }
    template <int PPT, int CH>
    __device__ __forceinline__ void tick() const { run<CH * PPT, CH * PPT + PPT>(); }
    template <int A, int B>
    __device__ __forceinline__ void run() const {
        if constexpr (A < B && A < NP) { piece<A>(); run<A + 1, B>(); }
    }
};
// the `flush` object the objective helpers see: tick slot CH = pieces [CH * PPT, CH * PPT + PPT) of an ImgFlusher (PPT = 0: the
// image is not complete yet -- an earlier segment's ticks do nothing)
template <class F, int PPT>
struct ImgTicks {
    const F& f;
    template <int CH> __device__ __forceinline__ void chunk() const { if constexpr (PPT > 0) f.template tick<PPT, CH>(); }
    template <int A, int B> __device__ __forceinline__ void range() const { if constexpr (PPT > 0) f.template run<A * PPT, B * PPT>(); }
    template <int A> __device__ __forceinline__ void rest() const { if constexpr (PPT > 0) f.template run<A * PPT, F::NP>(); }
};
template <int W, class IO, int ROWS = TRK_WAVE>
__device__ __forceinline__ ImgFlusher<W, IO, ROWS> spec_make_img(IO* __restrict__ out, int64_t base, int rows, int lane, IO* img) {
    IO* dst = out ? out + base * W : nullptr;
    const unsigned long long g = (unsigned long long)reinterpret_cast<uintptr_t>(dst);
    const unsigned long long gu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)g);
    const bool fast = out != nullptr && rows == ROWS && (g & 15) == 0;
    return ImgFlusher<W, IO, ROWS>{img, reinterpret_cast<const trk_f4*>(img) + lane, (unsigned)lane * 16u, gu,
                                   __builtin_amdgcn_ballot_w64(fast), lane};      // a ballot is scalar by definition: all lanes, or none
}
// ragged last wavefront / unaligned view: the staged image is copied with plain stores (wave-uniform branch, after the last staging)
template <int W, class IO, int ROWS>
__device__ __forceinline__ void spec_img_copy_slow(const ImgFlusher<W, IO, ROWS>& f, IO* __restrict__ out, int64_t base, int rows) {
    if (out == nullptr || f.on != 0ull) return;
    for (int e = f.lane; e < rows * W; e += TRK_WAVE) out[base * W + e] = f.img[e];
}

// profiling hook: lane 0 of a wave records the shader clock at phase `k` (no-op when A.stamps == nullptr)
__device__ __forceinline__ void spec_stamp(unsigned long long* stamps, int64_t wblock, int k, int lane) {
    if (stamps) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) stamps[wblock * 8 + k] = t;
    }
}

// s_memtime counts shader clocks in a per-CU domain (tools/clock_calib.hip: 2.39 GHz, unsynchronised between CUs); the
// constant 100 MHz s_memrealtime is chip-wide, so one such stamp per wave places the waves on a common time axis.
__device__ __forceinline__ void spec_stamp_real(unsigned long long* stamps, int64_t wblock, int k, int lane) {
    if (stamps) {
        // bits 0..43 time, 44..59 HW_ID[15:0] (wave, simd, pipe, cu, sh, se), 60..63 XCC_ID -- where the wave ran
        const unsigned long long hw = (unsigned)__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffffu;
        const unsigned long long xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
        const unsigned long long t = (__builtin_amdgcn_s_memrealtime() & ((1ull << 44) - 1)) | (hw << 44) | (xcc << 60);
        if (lane == 0) stamps[wblock * 8 + k] = t;
    }
}

__device__ __forceinline__ float spec_wave_sum(float v) { return trk_wave_sum(v); }

// ---------------------------------------------------------------------------------------------------------
// collision objectives on NL link points held in registers.  Adds w * cost to `cost` and
// w * d cost / d p to (gx, gy, gz) (accumulating).  Margins are C.obj_link_margin[0..NL) in baked order.
// ---------------------------------------------------------------------------------------------------------
template <int NL, class Tick, bool FAST = false, bool GENERAL = true>
__device__ __forceinline__ float spec_objects_cost(const DevCostHdr& C, float w, const float (&px)[NL],
                                                   const float (&py)[NL], const float (&pz)[NL], float (&gx)[NL],
                                                   float (&gy)[NL], float (&gz)[NL], const Tick& tick, const float4* lds_spheres,
                                                   int mbase = 0, const float4* lds_prims = nullptr) {
    float s[NL], ax[NL], ay[NL], az[NL];
    scene_min_sdf<NL, const Tick&, FAST, GENERAL>(C, px, py, pz, s, ax, ay, az, tick, lds_spheres, lds_prims);
    float cost = 0.0f;
    if (C.clamp_fields & TRK_FIELD_OBJECTS) {                                  // wave-uniform: the hinge form (clamp_sdf=True)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float v = cptr(C.obj_link_margin)[mbase + l] - s[l];
            const float wl = v > 0.0f ? w : 0.0f;                              // relu: value and gradient vanish at or below zero
            cost += __builtin_fmaxf(v, 0.0f);
            gx[l] = fmaf(-wl, ax[l], gx[l]); gy[l] = fmaf(-wl, ay[l], gy[l]); gz[l] = fmaf(-wl, az[l], gz[l]);
        }
        return w * cost;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        cost += cptr(C.obj_link_margin)[mbase + l] - s[l];                     // sum_l max_o (margin - sdf_o) = sum_l (margin - min_o sdf_o)
        gx[l] = fmaf(-w, ax[l], gx[l]); gy[l] = fmaf(-w, ay[l], gy[l]); gz[l] = fmaf(-w, az[l], gz[l]);
    }
    return w * cost;
}

// Arm-per-lane kernels (k_rollout_gpa): lanes 2s / 2s + 1 of a wavefront hold the two ISOMORPHIC arms of sample s, each with the NL
// collision links of ITS arm; the margins of arm a are C.obj_link_margin[a * NL .. a * NL + NL) -- two scalar loads and a select per link.
template <int NL, class Tick, bool FAST = false, bool GENERAL = true>
__device__ __forceinline__ float spec_objects_cost_arm(const DevCostHdr& C, float w, const float (&px)[NL], const float (&py)[NL],
                                                       const float (&pz)[NL], float (&gx)[NL], float (&gy)[NL], float (&gz)[NL],
                                                       const Tick& tick, const float4* lds_spheres, bool odd, const float4* lds_prims = nullptr) {
    float s[NL], ax[NL], ay[NL], az[NL];
    scene_min_sdf<NL, const Tick&, FAST, GENERAL>(C, px, py, pz, s, ax, ay, az, tick, lds_spheres, lds_prims);
    float cost = 0.0f;
    const bool hinge = (C.clamp_fields & TRK_FIELD_OBJECTS) != 0;             // wave-uniform: clamp_sdf=True
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const float m0 = cptr(C.obj_link_margin)[l], m1 = cptr(C.obj_link_margin)[NL + l];
        const float v = (odd ? m1 : m0) - s[l];
        const float wl = (hinge && !(v > 0.0f)) ? 0.0f : w;                     // relu: value and gradient vanish at or below zero
        cost += hinge ? __builtin_fmaxf(v, 0.0f) : v;
        gx[l] = fmaf(-wl, ax[l], gx[l]); gy[l] = fmaf(-wl, ay[l], gy[l]); gz[l] = fmaf(-wl, az[l], gz[l]);
    }
    return w * cost;
}
template <int NL>
__device__ __forceinline__ float spec_ws_cost_arm(const DevCostHdr& C, float w, const float (&px)[NL], const float (&py)[NL],
                                                  const float (&pz)[NL], float (&gx)[NL], float (&gy)[NL], float (&gz)[NL], bool odd) {
    float cost = 0.0f;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const float m0 = cptr(C.obj_link_margin)[l], m1 = cptr(C.obj_link_margin)[NL + l];
        cost += ws_cost_point(C, odd ? m1 : m0, px[l], py[l], pz[l], w, gx[l], gy[l], gz[l]);
    }
    return w * cost;
}
// the partner lane of an arm pair (lane ^ 1) as a DPP operand: quad_perm:[1,0,3,2]
__device__ __forceinline__ float trk_dpp_partner(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}

// the object field's cost of ONE collision link at a wave-uniform (constant) position: w * (margin - sdf) [hinge if clamped]; no gradient
__device__ __forceinline__ float spec_object_cost_uniform_point(const DevCostHdr& C, float w, float x, float y, float z, int m_index, int lane,
                                                                const float4* lds_prims = nullptr) {
    const float v = cptr(C.obj_link_margin)[m_index] - scene_min_sdf_uniform_point(C, x, y, z, lane, lds_prims);
    return w * ((C.clamp_fields & TRK_FIELD_OBJECTS) ? __builtin_fmaxf(v, 0.0f) : v);
}

// ---------------------------------------------------------------------------------------------------------
// boolean collision fields on NL link points held in registers (distance_fields.py:210-215, 283-291; tasks.py:227-228 ORs the
// fields).  Objects: the scene's minimum signed distance comes from the same ranking as the cost (one rsq per point); a
// lane whose distance lies within 1e-5 of its margin -- where the last-ulp difference between n2 * rsq(n2) and the
// reference's sqrt could flip the comparison -- re-evaluates that point with IEEE sqrt object by object, so the byte equals
// the table-driven kernel's (and the oracle's) on every input.
// ---------------------------------------------------------------------------------------------------------
#ifndef TRK_COLL_FAST_SPHERES
#define TRK_COLL_FAST_SPHERES 1      // 0: experiment / A-B -- every scene takes the ranking path with the arg-min index
#endif
template <int NL>
__device__ __forceinline__ bool spec_collision_links(const DevCostHdr& C, int fields, float margin, int use_default,
                                                     const float (&px)[NL], const float (&py)[NL], const float (&pz)[NL],
                                                     const float4* lds_spheres, int mbase = 0) {
    bool hit = false;
    float mg[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) mg[l] = use_default ? cptr(C.obj_link_margin)[mbase + l] : margin;
    if ((fields & TRK_FIELD_OBJECTS) && C.n_objects > 0 && scene_is_fast(C) && TRK_COLL_FAST_SPHERES) {
        // Round 6 -- a scene of <= 16 spheres of one radius r and nothing else (wave-uniform): the boolean needs the nearest sphere's
        // DISTANCE, not the sphere.  min_c |p - c|^2 = min_c (p.(-2c) + |c|^2) + |p|^2: the ranking keys without an index riding in
        // their mantissas (no v_and_or per sphere and point), no gather of the winner's centre, no second evaluation -- and the test
        // d < margin + r in squared form.  A lane within 1e-5 of its threshold (in d: |d^2 - thr^2| = (d + thr) |d - thr|) re-evaluates
        // object by object with IEEE sqrt exactly like the general path below, so the byte is the same on every input.
        float bk[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) bk[l] = __builtin_inff();
        const TRK_CAS float* tab = cptr(C.sphere_pairs);
        for (int j = 0; j < C.n_sphere_pairs; ++j) {
            const F8 rec = load_f8_uniform(tab, j);
            const trk_f2 cx = {rec.v[0], rec.v[1]}, cy = {rec.v[2], rec.v[3]}, cz = {rec.v[4], rec.v[5]}, cw = {rec.v[6], rec.v[7]};
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const trk_f2 key = __builtin_elementwise_fma(trk_f2{px[l], px[l]}, cx,
                                   __builtin_elementwise_fma(trk_f2{py[l], py[l]}, cy,
                                   __builtin_elementwise_fma(trk_f2{pz[l], pz[l]}, cz, cw)));
                bk[l] = __builtin_fminf(bk[l], __builtin_fminf(key.x, key.y));
            }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float thr = mg[l] + C.sphere_r;                          // wave-uniform
            const float d2 = fmaf(px[l], px[l], fmaf(py[l], py[l], fmaf(pz[l], pz[l], bk[l])));
            bool h = thr > 0.0f && d2 < thr * thr;
            if (__builtin_fabsf(d2 - thr * thr) < 2.5e-5f * __builtin_fmaxf(thr, 0.04f) || !(thr > 1e-4f)) {      // rare: exact, object by object
                h = false;
                for (int o = 0; o < C.n_objects; ++o) {
                    float gx, gy, gz;
                    h |= object_sdf<true>(C, o, px[l], py[l], pz[l], gx, gy, gz) < mg[l];
                }
            }
            hit |= h;
        }
    } else if ((fields & TRK_FIELD_OBJECTS) && C.n_objects > 0) {
        float s[NL], ax[NL], ay[NL], az[NL];
        NoTick nt;
        scene_min_sdf<NL, const NoTick&, false>(C, px, py, pz, s, ax, ay, az, nt, lds_spheres);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            bool h = s[l] < mg[l];
            if (__builtin_fabsf(s[l] - mg[l]) < 1e-5f) {            // rare: exact re-evaluation, object by object
                h = false;
                for (int o = 0; o < C.n_objects; ++o) {
                    float gx, gy, gz;
                    h |= object_sdf<true>(C, o, px[l], py[l], pz[l], gx, gy, gz) < mg[l];
                }
            }
            hit |= h;
        }
    }
    if ((fields & TRK_FIELD_WS) && C.has_ws) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
            hit |= (px[l] - C.ws_min[0] < mg[l]) | (py[l] - C.ws_min[1] < mg[l]) | (pz[l] - C.ws_min[2] < mg[l]) |
                   (C.ws_max[0] - px[l] < mg[l]) | (C.ws_max[1] - py[l] < mg[l]) | (C.ws_max[2] - pz[l] < mg[l]);
    }
    return hit;
}
// the same pair for the FACTORISED accumulation of the attached-point kernels: adds w * (margin - ||pa - pb||) to `cost` and returns
// s = w / ||pa - pb|| (0 at coincident points / behind the hinge) -- the force on a is sum_pairs s (pb - pa), on b the opposite
__device__ __forceinline__ float spec_self_pair_s(float w, float margin, float ax, float ay, float az, float bx, float by, float bz,
                                                  bool clamp, float& cost) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rs = n2 > 0.0f ? trk_rsq(n2) : 0.0f;
    const float nrm = n2 * rs;
    if (clamp) w = margin - nrm > 0.0f ? w : 0.0f;
    cost = fmaf(w, margin - nrm, cost);
    return w * rs;
}
// the same, handing back the difference d = pa - pb as well: the force on a is -s d, on b +s d.  The attached-point kernels accumulate THESE
// per point (one FMA per component and side).  The first factorised form accumulated S = sum s and V = sum s pb and formed V - pa S at the
// end -- fewer instructions when components are constants, but s (pb - pa) computed as s pb - s pa cancels catastrophically when the two
// points are close and far from the origin (two points 1 mm apart on one link: relative error 1e-7 |p| / |d| = 1e-4 of the force, found by
// the point-set fuzz with more seeds); differences first is translation invariant like the reference's own expression.
__device__ __forceinline__ float spec_self_pair_sd(float w, float margin, float ax, float ay, float az, float bx, float by, float bz,
                                                   bool clamp, float& cost, float& dx, float& dy, float& dz) {
    dx = ax - bx; dy = ay - by; dz = az - bz;
    const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rs = n2 > 0.0f ? trk_rsq(n2) : 0.0f;
    const float nrm = n2 * rs;
    if (clamp) w = margin - nrm > 0.0f ? w : 0.0f;
    cost = fmaf(w, margin - nrm, cost);
    return w * rs;
}
// one self-collision pair, boolean (distance_fields.py:210-215): ||pa - pb|| < margin, IEEE sqrt like torch.linalg.norm
__device__ __forceinline__ bool spec_self_hit(float margin, float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz))) < margin;
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
                                                float& gbz, bool clamp = false) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    const float n2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rs = n2 > 0.0f ? trk_rsq(n2) : 0.0f;              // one transcendental: 1/||d|| (0 at d = 0, like torch.norm's backward)
    const float nrm = n2 * rs;
    if (clamp) w = margin - nrm > 0.0f ? w : 0.0f;                // relu(margin - d): value and gradient vanish at or below zero
    const float inv = w * rs;
    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
    gax -= ux; gay -= uy; gaz -= uz;
    gbx += ux; gby += uy; gbz += uz;
    return w * (margin - nrm);
}
