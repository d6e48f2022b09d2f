// trk_launch.h -- host-callable launchers defined next to the kernels they start.
#pragma once
#include "trk_device.h"

void trk_launch_fk_forward(int mode, const DevModelHdr& hdr, const DevLink* links, const SelMap& sel, int n_sel,
                           const float* q, int64_t n, float* out, hipStream_t st);
// selp = sel re-indexed by walk position (selp.col[p] = sel.col[link at position p])
void trk_launch_fk_backward(int mode, const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const SelMap& sel,
                            const SelMap& selp, int n_sel, const float* q, const float* gin, int64_t n, float* gq, hipStream_t st);
void trk_launch_ik_step(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, int link, const float* H_target,
                        int per_sample, const float* lower, const float* upper, float w_jl, float se3_eps, float lr,
                        const IkSchedule& sched, int n_steps, int64_t n, float* q, float* mom, float* vel, float* loss,
                        uint8_t* valid, hipStream_t st);
void trk_launch_cost_fields(const DevCostHdr& C, int fields, const float* link_pos, int64_t n, const float* gcost,
                            float* cost, float* g_link_pos, hipStream_t st);
void trk_launch_collision_fields(const DevCostHdr& C, int fields, const float* link_pos, int64_t n, float margin,
                                 int use_default, uint8_t* out, hipStream_t st);
void trk_launch_ee_cost(const DevCostHdr& C, const float* H, int64_t n, int64_t stride, const float* target, int per_sample,
                        const float* gcost, float* cost, float* gH, int64_t g_stride, hipStream_t st);
// ps == nullptr: the cost model's columns are the links; otherwise the attached points of *ps
// io_mode: 0 fp32 | 1 q / link_pos / gq are _Float16 in HBM | 2 q / link_pos _Float16, gq fp32; arithmetic, cost and cost_sum are always
// fp32.  fp16 q: the gradient is multiplied by grad_scale before the store, fp16 gradient stores saturate at +-65504.
void trk_launch_rollout_generic(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const DevPointSet* ps,
                                const DevCostHdr& C, const TrkRolloutWeights& w, int io_mode, float grad_scale, const void* q, int64_t n,
                                void* link_pos, float* cost, void* gq, float* cost_sum, hipStream_t st);
void trk_launch_fk_points(const DevModelHdr& hdr, const DevLink* links, const DevPointSet& ps, const float* q, int64_t n,
                          float* out, hipStream_t st);
void trk_launch_fk_points_backward(const DevModelHdr& hdr, const DevLink* links, const int32_t* fin, const DevPointSet& ps,
                                   const float* q, const float* gin, int64_t n, float* gq, hipStream_t st);
size_t trk_lds_rollout(const DevModelHdr& hdr, int n_cols);                 // dynamic LDS bytes a launch needs
size_t trk_lds_fk_points(const DevModelHdr& hdr, int n_points, bool backward);
void trk_launch_fk_jacobian(const DevModelHdr& hdr, const DevLink* links_dev, const DevLink* links_host, const float* q,
                            const float* qd, int64_t n, int link, int link_joint_idx, float* pos, float* quat,
                            float* lin_jac, float* ang_jac, float* vel_lin, float* vel_ang, hipStream_t st);
void trk_launch_fk_analytic_jacobian(const DevModelHdr& hdr, const DevLink* links, const void* dofs, const float* q,
                                     int64_t n, float* J, hipStream_t st);
void trk_launch_rotmat_to_quat(const float* R, int64_t n, int stride, int pitch, float* out, hipStream_t st);
void trk_launch_frame_compose(int op, const float* Ra, const float* ta, int a_bcast, const float* Rb, const float* tb, int b_bcast,
                              int64_t n, float* Ro, float* to, hipStream_t st);
void trk_launch_frame_compose_bwd(int op, const float* Ra, const float* ta, const float* Rb, const float* tb, const float* gR,
                                  const float* gt, int64_t n, float* gRa, float* gta, float* gRb, float* gtb, hipStream_t st);
void trk_launch_frame_transform_points(const float* R, const float* t, int64_t n, const float* pts, int P, float* out,
                                       hipStream_t st);
void trk_launch_frame_transform_points_bwd(const float* g, int64_t n, const float* pts, int P, float* gR, float* gt,
                                           hipStream_t st);
void trk_launch_frame_quat_euler(const float* R, int64_t n, int stride, int pitch, float* quat_xyzw, float* euler,
                                 hipStream_t st);
void trk_launch_frame_quat_euler_bwd(const float* R, int64_t n, int stride, int pitch, const float* gquat_xyzw, const float* geuler,
                                     float* gR, hipStream_t st);
void trk_launch_rotation_from(int axis, const float* in, int64_t n, float* R, const float* gR, float* gin, hipStream_t st);
void trk_launch_grid_precompute(const DevCostHdr& C, const int32_t* dims, const float* lo, const float* hi, float* sdf,
                                float* grad, hipStream_t st);
void trk_launch_traj_validate(const uint8_t* wp, const float* x, int64_t T, int H, int S, int Hi, int D, const float* qmin,
                              const float* qmax, int64_t inner, uint8_t* flags, int64_t* idx, int32_t* counts,
                              int32_t* counts_host, int32_t ticket, float* gathered, hipStream_t st);
void trk_launch_grid_pack(const float* sdf, const float* grad, const int32_t dims[3], int nb1, int nb2, int64_t n_rec, float4* cells, hipStream_t st);
int trk_launch_jtj(int mfma, const float* lin, const float* ang, const float* r6, int64_t n, int D, float* JtJ, float* Jtr,
                   const float* damping, int damping_stride, float* dq, hipStream_t st);
size_t trk_pack_scratch_floats(int H, int D);
void trk_launch_pack_sums(const float* cost, const void* gq, int grad_f16, float unscale, const float* block_sums, const float* traj_cost,
                          int B, int H, int D, int64_t nb, float* scratch, float* out, hipStream_t st);
void trk_launch_scale_rows(int f16, const void* g, const float* sc, int sc_stride, int64_t n, int D, void* out, hipStream_t st);
void trk_launch_interpolate_columns(const float* x, int64_t n, int L, int C, int K, const int32_t* src, const float* w, float* out,
                                    hipStream_t st);
void trk_launch_interpolate_columns_bwd(const float* g, int64_t n, int L, int C, int K, const int32_t* src, const float* w, float* gx,
                                        hipStream_t st);
void trk_launch_interpolate(const float* x, int64_t T, int H, int D, int n_interp, const float* alpha, const float* beta,
                            float* out, hipStream_t st);
// returns -1: a trajectory does not fit the LDS, -2: fp32 trajectories with fp16 gradients (not a mode)
int trk_launch_gp_prior(int f16, int grad_f16, float grad_scale, const void* q, const void* qd, int64_t B, int H, int D, float dt, float sigma,
                        float w, float* cost, void* gq, void* gqd, int accumulate, hipStream_t st);
void trk_launch_gp_sample_cost(int f16, const void* q, const void* qd, int64_t n, int H, int D, float dt, float sigma, float w, float* cost,
                               float* block_sums, hipStream_t st);
void trk_launch_finite_difference(const float* x, int64_t B, int H, int D, float dt, int method, float* out, hipStream_t st);
void trk_launch_traj_diff_norm_sum(const float* x, int64_t B, int H, int S, int c0, int D, float* out, hipStream_t st);
void trk_launch_reduce_sum(const float* x, int64_t n, float* out, hipStream_t st);
void trk_launch_sdf_points(const DevCostHdr& C, const float* pts, int64_t n, float* sdf, float* grad, hipStream_t st);
// trajectories that 64 consecutive samples can touch when a trajectory has `hi` samples: floor((hi - 1 + 63) / hi) + 1 (SpecArgs::via_slots)
inline int trk_via_slots(int64_t hi) { return (int)((hi + 62) / hi) + 1; }
// raises the dynamic-LDS ceiling of every kernel once (gfx950: 160 KiB per workgroup)
int trk_kernels_init(void);
// trk_capi.hip's error string / initialisation, for the other translation units of the library
int trk_fail(int code, const char* msg);
int trk_hip_fail(int hip_error, const char* what);
int trk_ensure_init(void);
