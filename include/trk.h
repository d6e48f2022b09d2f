/*
 * trk.h -- C ABI of libtrk.so: MI355X (gfx950) batched differentiable forward
 * kinematics + planning-objective kernels.
 *
 * This is the drop-in boundary for the hot path of anindex/torch_robotics
 * (SURVEY.md section 8b).  The reference has no FFI layer: its boundary is a set of
 * Python methods.  Each entry point below names the reference method whose
 * arithmetic it replaces (file:line relative to the reference repo); the Python
 * classes in torch_robotics_amd/ keep the reference's signatures and call these.
 *
 * Conventions
 *  - All tensor arguments are DEVICE pointers to contiguous row-major fp32 unless
 *    a comment says "host".  The caller allocates every input and output; the
 *    library owns only the opaque handles it returns.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launches
 *    are asynchronous; the caller synchronises.  No hidden global state; handles
 *    are immutable except through the trk_*_set_* calls, which must not race with
 *    launches that use the handle.
 *  - Every function returns TRK_OK (0) or a negative TrkStatus and writes nothing
 *    on error.  Nothing throws or aborts across the boundary.
 *  - N = number of samples (batch x horizon folded, reference robot_panda.py:142),
 *    L = number of links (URDF file order), D = number of DOFs (file order of the
 *    non-fixed links, reference robot_tree.py:112-115).
 *  - Homogeneous transforms are 4x4 row-major with the translation in column 3 and
 *    bottom row [0,0,0,1] (reference frame.py:81-85).  Quaternions are wxyz.
 */
#ifndef TRK_H
#define TRK_H

#ifndef __HIPCC_RTC__       /* in-process device compilation of a generated unit: the integer types are built in */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define TRK_ABI_VERSION 5
#define TRK_MAX_LINKS 64
#define TRK_MAX_DOFS 32
#define TRK_MAX_POSE_SLOTS 8
#define TRK_MAX_OBJECTS 16
#define TRK_MAX_PRIMS 256
#define TRK_MAX_COLL_LINKS 192
#define TRK_MAX_SELF_PAIRS 1024
#define TRK_MAX_VIRTUAL 192

typedef enum TrkStatus {
    TRK_OK = 0,
    TRK_ERR_INVALID_ARG = -1,   /* null pointer, negative size, index out of range          */
    TRK_ERR_UNSUPPORTED = -2,   /* joint type / model size the kernels do not handle         */
    TRK_ERR_HIP = -3,           /* a HIP runtime call failed (see trk_last_error)            */
    TRK_ERR_NO_DEVICE = -4      /* no gfx950 device visible                                  */
} TrkStatus;

typedef enum TrkJointType {
    TRK_JOINT_FIXED = 0,
    TRK_JOINT_REVOLUTE = 1,
    TRK_JOINT_CONTINUOUS = 2,
    TRK_JOINT_PRISMATIC = 3,
    TRK_JOINT_UNSUPPORTED = 4
} TrkJointType;

typedef void* trk_stream_t;

/* ---------------------------------------------------------------------------------
 * Kinematic model.  Host arrays, copied once at create().  Produced by
 * torch_robotics_amd/kinmodel.py, which restates the reference's model build
 * (models/utils.py:199-313, rigid_body.py:74-123, robot_tree.py:77-126).
 * All per-link arrays are indexed by link (file order) unless marked "by position"
 * (= index into the DFS pre-order `order`).
 * --------------------------------------------------------------------------------- */
typedef struct TrkKinModelDesc {
    int32_t abi_version;        /* TRK_ABI_VERSION */
    int32_t n_links;            /* L <= TRK_MAX_LINKS */
    int32_t n_dofs;             /* D <= TRK_MAX_DOFS  */
    int32_t n_slots;            /* pose-stack slots used by parent_slot/store_slot */
    const int32_t* parent;      /* [L] parent link, -1 for the root (link 0) */
    const int32_t* joint_type;  /* [L] TrkJointType of the joint whose child is the link */
    const int32_t* dof_idx;     /* [L] DOF index or -1 */
    const float* R_fixed;       /* [L*9] Rz(yaw)Ry(pitch)Rx(roll), row-major (rigid_body.py:93) */
    const float* trans;         /* [L*3] joint origin xyz */
    const float* axis;          /* [L*3] raw <axis>, zeros when absent */
    const int32_t* rot_axis;    /* [L] stateless FK rotation axis 0/1/2 (rigid_body.py:163-168) */
    const float* rot_sign;      /* [L] sign(axis_k) in {-1,0,+1} */
    const int32_t* clamp;       /* [L] stateless FK clamps q to [lower,upper] (rigid_body.py:157-160) */
    const float* lower;         /* [L] */
    const float* upper;         /* [L] */
    const int32_t* sf_rot_axis; /* [L] stateful path axis, sign ignored (rigid_body.py:101-106,226-227) */
    const int32_t* sf_clamp;    /* [L] stateful path clamps whenever limits exist (rigid_body.py:218-224) */
    const int32_t* jac_axis;    /* [L] column of the world rotation used as joint axis (robot_tree.py:242-244) */
    const int32_t* joint_list_idx; /* [L] index of the link's joint in the URDF joint list (robot_tree.py:235-240) */
    const int32_t* order;       /* [L] DFS pre-order of links, order[0] == 0 */
    const int32_t* subtree_end; /* [L] by position: one past the last descendant's position */
    const int32_t* parent_slot; /* [L] by position: -1 = parent is the previous position, else slot to load */
    const int32_t* store_slot;  /* [L] by position: slot to keep this link's pose in, or -1 */
    float base_R[9];            /* root pose (robot_tree.py:133-134), identity by default */
    float base_t[3];
} TrkKinModelDesc;

typedef struct TrkModel TrkModel;

int trk_abi_version(void);
/* Human-readable text of the last error on the calling thread ("" if none). */
const char* trk_last_error(void);

int trk_model_create(const TrkKinModelDesc* desc, TrkModel** out);
void trk_model_destroy(TrkModel* model);
/* reference: DifferentiableTree.update_base_pose robot_tree.py:133-134 (pose already converted
 * to a rotation matrix on the host, quaternion.py:102-120).  Host pointers. */
int trk_model_set_base_pose(TrkModel* model, const float* R9, const float* t3);
int trk_model_n_links(const TrkModel* model);
int trk_model_n_dofs(const TrkModel* model);
/* 1 if a model-specialised (ahead-of-time unrolled) fused kernel exists for this model and is enabled. */
int trk_model_is_specialized(const TrkModel* model);
/* Switch between the model-specialised fused kernel (default when one was built for these tables) and the
 * table-driven one; both compute the same function (used by the parity tests to cover both). */
int trk_model_enable_specialized(TrkModel* model, int enable);
/* Number of model-specialised units registered with the library (ahead-of-time ones plus run-time compiled ones that were
 * dlopen'ed).  A unit compiled against another layout of the launch structures is refused at registration and not counted;
 * the run-time compiler (torch_robotics_amd/jit.py) uses this to detect -- and rebuild -- a stale on-disk unit. */
int trk_spec_count(void);

/* ---------------------------------------------------------------------------------
 * Forward kinematics (stateless path).
 * --------------------------------------------------------------------------------- */

/* reference: DifferentiableTree.compute_forward_kinematics_all_links robot_tree.py:267-301
 *            (+ DifferentiableRigidBody.forward_kinematics rigid_body.py:146-211,
 *               Frame.get_transform_matrix frame.py:81-85).
 * q [N,D] -> H_out [N, n_sel, 4, 4].  link_sel: host int32[n_sel] link indices in output
 * order (NULL = all L links in file order, n_sel ignored). */
int trk_fk_forward(const TrkModel* model, const float* q, int64_t n,
                   const int32_t* link_sel, int32_t n_sel, float* H_out, trk_stream_t stream);

/* Link origins only.  reference: RobotPanda.fk_map_collision_impl robot_panda.py:138-170
 * + link_pos_from_link_tensor geometrics/utils.py:321-328.
 * q [N,D] -> pos_out [N, n_sel, 3]. */
int trk_fk_positions(const TrkModel* model, const float* q, int64_t n,
                     const int32_t* link_sel, int32_t n_sel, float* pos_out, trk_stream_t stream);

/* Explicit reverse-mode of trk_fk_forward (replaces autograd through ~350 bmm nodes).
 * gH [N, n_sel, 4, 4] (bottom row ignored) -> gq [N,D]; zero where the reference clamps
 * (rigid_body.py:157-160). */
int trk_fk_backward(const TrkModel* model, const float* q, const float* gH, int64_t n,
                    const int32_t* link_sel, int32_t n_sel, float* gq, trk_stream_t stream);

/* Explicit reverse-mode of trk_fk_positions.  gpos [N, n_sel, 3] -> gq [N,D]. */
int trk_fk_positions_backward(const TrkModel* model, const float* q, const float* gpos, int64_t n,
                              const int32_t* link_sel, int32_t n_sel, float* gq, trk_stream_t stream);

/* ---- points rigidly attached to links ----------------------------------------------------------------------------
 * reference: RobotPanda.fk_map_collision_impl robot_panda.py:154-168 (grasped-object collision points,
 *            GraspedObjectPandaBox.get_base_points_for_collision objects.py:57-89) through Frame.transform_point
 *            frame.py:116-118:  world = R_link * offset + t_link.  A zero offset is the link origin, so a point set
 *            {all links, then the grasped points} is exactly what fk_map_collision returns: [N, L + G, 3].
 *            Also carries per-link collision spheres (data/configs/panda/panda_sphere_config.yaml).
 * point_link: host int32[n_points] link (file index) each point is fixed to; point_offset: host float[n_points*3]
 * in that link's frame.  Output column k is point k.  The handle owns small device tables; immutable afterwards. */
typedef struct TrkPointSet TrkPointSet;
#define TRK_MAX_POINTS 192
int trk_point_set_create(const TrkModel* model, const int32_t* point_link, const float* point_offset,
                         int32_t n_points, TrkPointSet** out);
void trk_point_set_destroy(TrkPointSet* ps);
int trk_point_set_size(const TrkPointSet* ps);
/* 1 if a generated fused kernel with exactly this point set baked in was built (see trk_model_is_specialized). */
int trk_point_set_is_specialized(const TrkPointSet* ps);

/* q [N,D] -> pos_out [N, n_points, 3]. */
int trk_fk_points(const TrkModel* model, const TrkPointSet* ps, const float* q, int64_t n, float* pos_out,
                  trk_stream_t stream);
/* Explicit reverse mode of trk_fk_points (replaces autograd through the FK recursion and transform_point):
 * gpos [N, n_points, 3] -> gq [N,D]; zero where the reference clamps (rigid_body.py:157-160). */
int trk_fk_points_backward(const TrkModel* model, const TrkPointSet* ps, const float* q, const float* gpos, int64_t n,
                           float* gq, trk_stream_t stream);

/* Stateful FK + geometric Jacobian of one link.
 * reference: DifferentiableTree.compute_forward_kinematics_and_geometric_jacobian
 *            robot_tree.py:218-248 (update_kinematic_state :136-190, Frame.get_quaternion
 *            frame.py:87-114, q_convert_wxyz quaternion.py:240-242).
 * q, qd [N,D] -> pos [N,3], quat_wxyz [N,4], lin_jac [N,3,D], ang_jac [N,3,D].
 * qd may be NULL (velocities are then not computed).  vel_lin / vel_ang [N,3] (the link's
 * spatial velocity in the link frame, spatial_vector.py:87-97) may be NULL. */
int trk_fk_jacobian(const TrkModel* model, const float* q, const float* qd, int64_t n, int32_t link,
                    float* pos, float* quat_wxyz, float* lin_jac, float* ang_jac,
                    float* vel_lin, float* vel_ang, trk_stream_t stream);

/* reference: DifferentiableTree.compute_analytical_jacobian_all_links robot_tree.py:250-265 (autograd Jacobian of
 * [pos(3), quat_wxyz(4)] of every link through the stateless FK and rotation_matrix_to_q).
 * q [N,D] -> J [N, L, 7, D]. */
int trk_fk_analytic_jacobian(const TrkModel* model, const float* q, int64_t n, float* J, trk_stream_t stream);

/* One iteration of the batched Adam IK, reference: DifferentiableTree.inverse_kinematics robot_tree.py:345-377 with
 * loss_fn_ik_per_q :386-417 (SE3_distance w_pos = w_rot = 1 + w_joint_limits * squared hinge on [lower, upper]) and
 * ik_termination :419-442.  In place: q [N,D] <- Adam(q, d loss / d q) with torch.optim.Adam's defaults (betas 0.9/0.999,
 * eps 1e-8), state adam_m / adam_v [N,D] (zero before step 1), step = 1-based iteration.  loss [N] and valid [N] (uint8:
 * inside the limits and SE3 error < se3_eps) are evaluated on q BEFORE the update, as in the reference loop; lr = 0 only
 * evaluates.  H_target: DEVICE [16] or [N,16]; lower / upper: DEVICE [D].  All nullable outputs may be NULL. */
int trk_ik_step(const TrkModel* model, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                const float* upper, float w_joint_limits, float se3_eps, float lr, int32_t step, int64_t n, float* q,
                float* adam_m, float* adam_v, float* loss, uint8_t* valid, trk_stream_t stream);

/* The same for n_steps consecutive iterations first_step, first_step + 1, ... in ONE launch per 32 iterations: the
 * configurations stay on the chip between iterations (the persistent form of the loop robot_tree.py:345-377).  Exactly
 * n_steps calls of trk_ik_step, except that loss / valid describe q as passed in (before the FIRST of the updates) -- a
 * caller that tests the termination condition every n_steps iterations loses nothing. */
int trk_ik_steps(const TrkModel* model, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                 const float* upper, float w_joint_limits, float se3_eps, float lr, int32_t first_step, int32_t n_steps,
                 int64_t n, float* q, float* adam_m, float* adam_v, float* loss, uint8_t* valid, trk_stream_t stream);

/* K damped Gauss-Newton (Levenberg-Marquardt) iterations of the batched IK in ONE launch -- BUILD-DEFINED: the reference's loop is
 * Adam (DifferentiableTree.inverse_kinematics robot_tree.py:303-384, served by trk_ik_steps); a Newton step is what its geometric
 * Jacobian (compute_forward_kinematics_and_geometric_jacobian robot_tree.py:218-248) is for.  Per sample and iteration:
 *   stateful FK + geometric Jacobian J (6 x D, the reference's column rule) of `link` at q;
 *   r = [p* - p ; rotation vector of R* R^T]  (world frame);   lambda = damping + lm_gain |r|^2;
 *   (J^T J + lambda I) dq = J^T r  (Cholesky);   q <- clamp(q + step_scale dq, lower, upper).
 * The Jacobian, the normal equations and their factor never leave the lane's registers (the two-launch form trk_fk_jacobian +
 * trk_jtj moves 416 bytes per sample and iteration).  H_target: DEVICE [16] (per_sample_target = 0) or [n,16]; lower / upper:
 * DEVICE [D]; q [n,D] in place; err [n] / valid [n] (nullable): SE3_distance(w_pos = w_rot = 1) of q AS PASSED IN and
 * `err < se3_eps and lower <= q <= upper` -- the metric and test of ik_termination (robot_tree.py:419-442), so that a caller
 * testing every K iterations sees what trk_ik_steps would report.  Checked against oracle/oracle_impl.inc: orc_ik_gn_step.
 * Served by generated kernels only (robots up to 9 DOF, the link a unit tracks): TRK_ERR_UNSUPPORTED otherwise. */
int trk_ik_gn_steps(const TrkModel* model, int32_t link, const float* H_target, int32_t per_sample_target, const float* lower,
                    const float* upper, float damping, float lm_gain, float step_scale, float se3_eps, int32_t n_steps, int64_t n,
                    float* q, float* err, uint8_t* valid, trk_stream_t stream);

/* reference: rotation_matrix_to_q quaternion.py:135-166 (via link_quat_from_link_tensor
 * geometrics/utils.py:341-344).  R: n matrices, `stride` floats apart, 3x3 block with row
 * pitch `row_pitch` (9/3 for packed rotations, 16/4 for 4x4 transforms).  -> quat_wxyz [n,4]. */
int trk_rotmat_to_quat(const float* R, int64_t n, int32_t stride, int32_t row_pitch,
                       float* quat_wxyz, trk_stream_t stream);

/* Rotation matrices from angles or quaternions: R [n,9] row-major.
 *   TRK_ROT_X / _Y / _Z   in = angle [n]      x_rot / y_rot / z_rot        geometrics/spatial_vector.py:8-47
 *   TRK_ROT_QUAT_WXYZ     in = quat  [n,4]    q_to_rotation_matrix         geometrics/quaternion.py:102-120
 * trk_rotation_from_backward: gR [n,9] -> gin, shaped like `in` ([n] angles, or [n,4] for the quaternion: the reference's
 * q_to_rotation_matrix is plain differentiable torch arithmetic, incl. the 2 / |q|^2 normalisation). */
#define TRK_ROT_X 0
#define TRK_ROT_Y 1
#define TRK_ROT_Z 2
#define TRK_ROT_QUAT_WXYZ 3
int trk_rotation_from(int32_t kind, const float* in, int64_t n, float* R_out, trk_stream_t stream);
int trk_rotation_from_backward(int32_t kind, const float* in, const float* gR, int64_t n, float* gin, trk_stream_t stream);

/* Frame algebra on packed poses: R [n,9] row-major, t [n,3] (reference: geometrics/frame.py:55-121, the `Frame`s that
 * compute_forward_kinematics_all_links(return_dict=True) returns, robot_tree.py:283-297).  A frame given once (na / nb == 1)
 * broadcasts against the other; n_out = max(na, nb).
 *   TRK_FRAME_COMPOSE      out = a o b       Frame.multiply_transform     frame.py:64-68, geometrics/utils.py:11-17
 *   TRK_FRAME_INVERSE      out = a^-1        Frame.inverse                frame.py:57-62  (b ignored)
 *   TRK_FRAME_INV_COMPOSE  out = b^-1 o a    Frame.multiply_inv_transform frame.py:70-76, geometrics/utils.py:21-29 */
#define TRK_FRAME_COMPOSE 0
#define TRK_FRAME_INVERSE 1
#define TRK_FRAME_INV_COMPOSE 2
int trk_frame_compose(int32_t op, const float* Ra, const float* ta, int64_t na, const float* Rb, const float* tb, int64_t nb,
                      float* R_out, float* t_out, trk_stream_t stream);
/* Reverse mode of the above for equal batch sizes (a broadcast input is expanded by the caller): (gR, gt) [n] ->
 * gRa, gta [n] and -- unless op is TRK_FRAME_INVERSE -- gRb, gtb [n]. */
int trk_frame_compose_backward(int32_t op, const float* Ra, const float* ta, const float* Rb, const float* tb, const float* gR,
                               const float* gt, int64_t n, float* gRa, float* gta, float* gRb, float* gtb, trk_stream_t stream);
/* Frame.transform_point frame.py:116-118: points [P,3] in the frames -> out [n,P,3] = R_s p + t_s (what fk_map_collision
 * does with a grasped object's points, robot_panda.py:154-168). */
int trk_frame_transform_points(const float* R, const float* t, int64_t n, const float* points, int32_t n_points, float* out,
                               trk_stream_t stream);
/* its reverse mode w.r.t. the poses: gout [n,P,3] -> gR [n,9] = sum_p g_p p^T, gt [n,3] = sum_p g_p */
int trk_frame_transform_points_backward(const float* gout, int64_t n, const float* points, int32_t n_points, float* gR,
                                        float* gt, trk_stream_t stream);
/* Frame.get_quaternion frame.py:87-114 (trace method; XYZW, the reference converts with q_convert_wxyz at
 * robot_tree.py:214-215) and Frame.get_euler frame.py:120-121 (roll, pitch, yaw).  R as in trk_rotmat_to_quat.
 * Either output may be NULL. */
int trk_frame_quat_euler(const float* R, int64_t n, int32_t stride, int32_t row_pitch, float* quat_xyzw, float* euler,
                         trk_stream_t stream);
/* Reverse mode of trk_frame_quat_euler w.r.t. the rotations: gquat_xyzw [n,4] and / or geuler [n,3] (either may be NULL)
 * -> gR [n,9] (packed).  Euler angles: the derivatives of torch.atan2 / torch.asin (frame.py:120-121).  Quaternion: the
 * reference multiplies its trace-method vector by the PYTHON float `0.5 / math.sqrt(tn * M[n,3,3])` (frame.py:112), which
 * autograd treats as a constant; the gradient here is the reference's: scale held fixed, branch as taken in the forward. */
int trk_frame_quat_euler_backward(const float* R, int64_t n, int32_t stride, int32_t row_pitch, const float* gquat_xyzw,
                                  const float* geuler, float* gR, trk_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Planning objectives ("cost model").  Host arrays, copied at create().
 * reference: torch_planning_objectives/fields/distance_fields.py, environments/primitives.py,
 *            environments/grid_map_sdf.py, robots/robot_base.py:105-141.
 * --------------------------------------------------------------------------------- */
typedef enum TrkPrimType {
    TRK_PRIM_SPHERE = 0,        /* MultiSphereField     primitives.py:108-112 */
    TRK_PRIM_ROUNDED_BOX = 1,   /* MultiBoxField        primitives.py:327-334 */
    TRK_PRIM_SHARP_BOX = 2      /* MultiSharpBoxField   primitives.py:220-223 */
} TrkPrimType;

typedef struct TrkPrimitive {
    int32_t type;               /* TrkPrimType */
    int32_t object;             /* index into objects[] */
    float center[3];
    float half[3];              /* boxes: sizes/2 */
    float radius;               /* sphere radius, or rounded-box rounding radius (0.15*min size) */
    float _pad;
} TrkPrimitive;

typedef struct TrkObject {      /* ObjectField pose, primitives.py:387-405: p' = R^T (p - pos) */
    float pos[3];
    float R[9];                 /* q_to_rotation_matrix(ori), row-major */
    int32_t prim_begin;         /* primitives [prim_begin, prim_end) belong to this object */
    int32_t prim_end;
    int32_t is_grid;            /* 1: this "object" is the precomputed grid (GridMapSDF) */
    int32_t _pad;
} TrkObject;

typedef struct TrkGridDesc {    /* GridMapSDF grid_map_sdf.py:9-117 (device pointers, caller-owned).  trk_cost_model_create packs
                                   the two arrays into its own (grad, sdf) records on the stream-0 timeline: like the host
                                   tables the grid is a snapshot taken at create(), later writes to sdf / grad are not seen */
    const float* sdf;           /* DEVICE [nx,ny,nz] */
    const float* grad;          /* DEVICE [nx,ny,nz,3] */
    int32_t dims[3];
    float lim_min[3];
    float map_dim[3];           /* |lim_max - lim_min| */
} TrkGridDesc;

typedef struct TrkCostModelDesc {
    int32_t abi_version;
    int32_t n_links_in;         /* links per sample in the position tensors handed to the cost ops */
    /* objects + workspace fields (CollisionObjectDistanceField / ...WorkspaceBoundaries...) */
    int32_t n_obj_links;        /* Lc */
    const int32_t* obj_link_idx;    /* [Lc] index into the position tensor */
    const float* obj_link_margin;   /* [Lc] collision_margins + cutoff_margin, added in fp32 on the host
                                       exactly as distance_fields.py:112 does */
    int32_t n_objects;
    const TrkObject* objects;
    int32_t n_prims;
    const TrkPrimitive* prims;
    int32_t has_grid;
    TrkGridDesc grid;
    int32_t has_ws;
    float ws_min[3];
    float ws_max[3];
    /* self collision (CollisionSelfField distance_fields.py:180-215) */
    int32_t n_self_links;       /* Ls */
    const int32_t* self_link_idx;   /* [Ls] index into the position tensor */
    int32_t n_self_pairs;       /* P */
    const int32_t* self_pairs;      /* [P*2] indices into self_link_idx.  A pair (a, a) is the reference's single-link case
                                       (distance_fields.py:195-198): "distance" 1e9 * |p_a|_1, so cost = margin - 1e9 |p_a|_1 */
    const float* self_margin;       /* [P] */
    /* end-effector SE(3) tracking (EESE3DistanceField distance_fields.py:335-359) */
    int32_t ee_link;            /* link whose pose is tracked (fused op); -1 = none */
    float ee_w_pos;
    float ee_w_rot;
    int32_t ee_square;
    float ee_target[16];        /* default target, row-major 4x4 */
    /* a second tracked link with its own target, same weights / square flag (two-arm scenes: BASELINE config 5 tracks
     * "EE on both arms"; the reference would sum two EESE3DistanceFields).  -1 = none. */
    int32_t ee2_link;
    float ee2_target[16];
    /* clamp_sdf=True of the fields (distance_fields.py:114-117): TRK_FIELD_* mask of the fields whose per-link (per-pair)
     * terms are relu(margin - signed distance) instead of margin - signed distance -- the hinge form planners optimise */
    int32_t clamp_fields;
    /* interpolate_link_pos=True of the embodiment fields (interpolate_points_v1 distance_fields.py:66-69, used at :145-147):
     * n_virtual extra position columns behind the n_links_in given ones,
     *     column n_links_in + k = virtual_w[2k] * column virtual_src[2k] + virtual_w[2k+1] * column virtual_src[2k+1],
     * which obj_link_idx / self_link_idx may name like any other column (0 <= index < n_links_in + n_virtual).  The table is
     * F.interpolate(mode='linear', align_corners=True) unrolled on the host: source columns and fp32 weights per output
     * point.  Position tensors handed to the cost ops keep n_links_in columns; gradients are scattered back to them. */
    int32_t n_virtual;              /* <= TRK_MAX_VIRTUAL */
    const int32_t* virtual_src;     /* [2*n_virtual] real columns (< n_links_in) */
    const float* virtual_w;         /* [2*n_virtual] */
} TrkCostModelDesc;

typedef struct TrkCostModel TrkCostModel;

int trk_cost_model_create(const TrkCostModelDesc* desc, TrkCostModel** out);
void trk_cost_model_destroy(TrkCostModel* cm);
/* reference: EESE3DistanceField.update_target distance_fields.py:344-345.  Host pointer, 16 floats. */
int trk_cost_model_set_ee_target(TrkCostModel* cm, const float* H16);
int trk_cost_model_set_ee2_target(TrkCostModel* cm, const float* H16);
/* trk_cost_fields may run a generated unit's field kernel when the cost model's columns and link sets equal a unit's collision
 * template (default: on); off = always the table-driven kernel (A/B tests). */
int trk_cost_model_enable_specialized(TrkCostModel* cm, int32_t on);

/* Which field a cost op evaluates. */
typedef enum TrkField {
    TRK_FIELD_SELF = 1,         /* CollisionSelfField */
    TRK_FIELD_OBJECTS = 2,      /* CollisionObjectDistanceField (analytic objects and/or grid) */
    TRK_FIELD_WS = 4            /* CollisionWorkspaceBoundariesDistanceField */
} TrkField;

/* reference: EmbodimentDistanceFieldBase.compute_embodiment_cost (field_type='sdf')
 *            distance_fields.py:107-124 for the fields in `fields` (bit-or of TrkField):
 *            cost[n] = sum over selected fields of  sum_links max_objects (margin - sdf).
 * link_pos [N, n_links_in, 3] -> cost [N].  g_link_pos (nullable) [N, n_links_in, 3] receives
 * d(sum_n gcost[n]*cost[n])/d link_pos (gcost NULL = ones): forward and explicit backward in
 * one pass, because the gradient is a by-product of the arg-min search. */
int trk_cost_fields(const TrkCostModel* cm, int32_t fields, const float* link_pos, int64_t n,
                    const float* gcost, float* cost, float* g_link_pos, trk_stream_t stream);

/* reference: compute_embodiment_collision (field_type='occupancy') distance_fields.py:210-215,
 *            283-291; PlanningTask ORs the fields tasks.py:227-228.
 * margin_override: NaN = per-link/per-pair default margins, else the scalar `margin=` kwarg.
 * -> in_collision [N] (uint8 0/1). */
int trk_collision_fields(const TrkCostModel* cm, int32_t fields, const float* link_pos, int64_t n,
                         float margin_override, uint8_t* in_collision, trk_stream_t stream);

/* reference: EESE3DistanceField.compute_costs_impl distance_fields.py:347-356 + SE3_distance
 *            geometrics/utils.py:130-154.
 * H_ee: n transforms `stride` floats apart (e.g. the last link of [N,L,4,4]: pointer offset
 * (L-1)*16, stride L*16).  target: DEVICE [16] (per_sample_target=0) or [N,16] (=1); NULL = the
 * cost model's stored target.  -> cost [N]; gH (nullable), same layout as H_ee, receives
 * gcost[n]*dcost/dH (gcost NULL = ones; all 16 floats of each transform are written, the constant
 * bottom row as zeros; other links' blocks of a strided gH are untouched). */
int trk_ee_cost(const TrkCostModel* cm, const float* H_ee, int64_t n, int64_t stride,
                const float* target, int32_t per_sample_target,
                const float* gcost, float* cost, float* gH, int64_t g_stride, trk_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Fused rollout: FK -> costs -> d cost / d q in one kernel (the north-star path).
 * reference chain replaced: PlanningTask.compute_collision_cost tasks.py:135-137,206-230
 *   (RobotPanda.fk_map_collision + the three collision fields) + EESE3DistanceField
 *   + `cost.sum().backward()`.
 * q [B,H,D] -> link_pos_out (nullable) [B,H,L,3], cost [B,H], gq [B,H,D];
 * cost = w_self*self + w_obj*objects + w_ws*workspace + w_ee*ee  (a zero weight skips the term);
 * gq = d cost[b,h] / d q[b,h,:] (samples are independent, so this equals the gradient of
 * cost.sum()).  cost_block_sums (nullable): DEVICE float[ceil(B*H/64)]; entry w receives the sum of
 * cost over samples [64w, 64w+64) -- one wave-level reduction and one plain store per wavefront, no
 * atomics, bit-reproducible.  With horizon 64 these are the per-trajectory costs; trk_reduce_sum folds
 * them into the scalar a multi-GPU caller all-reduces.
 * The cost model's link indices refer to the robot's L links (n_links_in == L).
 * --------------------------------------------------------------------------------- */
typedef struct TrkRolloutWeights {
    float w_self, w_obj, w_ws, w_ee;
} TrkRolloutWeights;

int trk_rollout_cost_grad(const TrkModel* model, const TrkCostModel* cm, const TrkRolloutWeights* w,
                          const float* q, int64_t batch, int32_t horizon,
                          float* link_pos_out, float* cost, float* gq, float* cost_block_sums,
                          trk_stream_t stream);

/* BASELINE config 4's step -- "FK + Jacobian + cost" -- as ONE call: trk_rollout_cost_grad plus the stateful FK + geometric Jacobian
 * of `link` (trk_fk_jacobian without velocities: pos [N,3], quat_wxyz [N,4], lin_jac / ang_jac [N,3,D], N = batch*horizon).
 * reference: PlanningTask._compute_collision_or_cost tasks.py:139-232 followed by
 * DifferentiableTree.compute_forward_kinematics_and_geometric_jacobian robot_tree.py:218-248 on the same q.
 * ONE launch when a generated unit serves the cost model, `link` is the unit's tracked link, link_pos_out is given and the stateful
 * walk coincides with the stateless one on the columns' chains (UR10 + Allegro: the Jacobian columns are read out of the poses the
 * rollout already holds -- no second walk, q read once); otherwise the two launches, same results (trk_last_dispatch tells). */
int trk_rollout_jacobian_cost_grad(const TrkModel* model, const TrkCostModel* cm, const TrkRolloutWeights* w, const float* q,
                                   int64_t batch, int32_t horizon, int32_t link, float* link_pos_out, float* cost, float* gq,
                                   float* cost_block_sums, float* pos, float* quat_wxyz, float* lin_jac, float* ang_jac,
                                   trk_stream_t stream);

/* 1 when trk_rollout_cost_grad(model, cm, w, ...) is served by a generated (model-specialised) kernel -- one whose baked
 * collision template equals the cost model's link sets for the terms `w` selects --, 0 when the table-driven kernel would run
 * (~10 x slower).  A deployment that must not fall back silently asserts this once after building its handles. */
int trk_rollout_is_specialized(const TrkModel* model, const TrkCostModel* cm, const TrkRolloutWeights* w);

/* Which kernel family served the CALLING THREAD's latest rollout call (trk_rollout_cost_grad[_f16], trk_rollout_gp_cost_grad,
 * trk_rollout_points_cost_grad, trk_rollout_collision[_via]).  No reference counterpart: the reference has one code path
 * (tasks.py:139-232); here a cost model whose link sets no generated unit bakes is served by the table-driven kernels, 10 - 30 x slower. */
enum {
    TRK_DISPATCH_NONE = 0,                  /* no rollout call yet on this thread (or a call that had nothing to launch) */
    TRK_DISPATCH_GENERATED = 1,             /* one launch of a generated (model-specialised) kernel */
    TRK_DISPATCH_TABLE = 2,                 /* the table-driven kernels */
    TRK_DISPATCH_GENERATED_PLUS_PRIOR = 3   /* trk_rollout_gp_cost_grad: generated rollout, the GP prior as launches of its own;
                                             * trk_rollout_jacobian_cost_grad: generated rollout, the Jacobian as a launch of its own */
};
int trk_last_dispatch(void);
/* trk_rollout_is_specialized for trk_rollout_points_cost_grad(ps->model, ps, cm, w, ...) (16-byte aligned point_pos_out assumed). */
int trk_rollout_points_is_specialized(const TrkPointSet* ps, const TrkCostModel* cm, const TrkRolloutWeights* w);
/* Strict mode (process-wide; also TRK_STRICT_SPECIALIZED=1 in the environment, read once): the rollout entry points above return
 * TRK_ERR_UNSUPPORTED instead of launching a table-driven kernel when the model HAS generated units but none matches the call.
 * Models without any generated unit are served as before.  Returns the previous setting. */
int trk_set_strict_specialized(int on);
/* A weight on a term the cost model does not have (w_self without self pairs, w_ws without a workspace box, w_obj on an empty
 * scene, w_ee without a tracked link) contributes nothing in either kernel family and never influences the dispatch. */

/* Registers a generated unit that was compiled to a CODE OBJECT in-process (torch_robotics_amd/jit.py: hipRTC, the fall-back when
 * no hipcc is installed) -- the unit's device half only; libtrk.so's generic launchers play its host half.  Used by the package's
 * run-time compiler, not by applications.  code: the code object (kept mapped by the HIP module); name_exprs / lowered_names
 * [n_kernels]: the kernels' C++ name expressions (as codegen.py lists them) and their mangled names; the *_ok / chunked /
 * fast_switch / jac_direct flags and the layout stamp come from the generator and the headers the object was compiled with
 * (a stamp of another layout is refused). */
typedef struct TrkModuleUnitDesc {
    int32_t spec_abi_version;
    uint32_t sizeof_args, sizeof_cost_hdr;
    const char* ident;
    uint64_t model_hash;
    int32_t n_links, n_dofs;
    int32_t n_obj_links; const int32_t* obj_link_idx;
    int32_t n_self_pairs; const int32_t* self_pairs;          /* [2 * n_self_pairs] LINK indices */
    int32_t ee_link, ee2_link;
    int32_t n_virtual; const int32_t* virtual_src; const float* virtual_w;
    int32_t chunked, fast_switch, fkhbwd_ok, fields_ok, ik_ok, ikgn_ok, jac_ok, jac_direct, gp_ok;
    const void* code; uint64_t code_size;
    int32_t n_kernels; const char* const* name_exprs; const char* const* lowered_names;
    /* n_points > 0: an ATTACHED-POINT unit (link spheres, grasped-object points): obj_link_idx / self_pairs are COLUMN indices of the
     * point set, points_hash identifies it (codegen.points_hash); its kernels are the fused rollout and the positions' reverse mode */
    int32_t n_points; uint64_t points_hash;
} TrkModuleUnitDesc;
int trk_spec_register_module(const TrkModuleUnitDesc* desc);
/* The layout stamp the generated units of this library carry (TRK_SPEC_ABI_VERSION, sizeof(SpecArgs) + sizeof(IkArgs) +
 * sizeof(IkGnArgs), sizeof(DevCostHdr)): what a TrkModuleUnitDesc must repeat.  out [3]. */
int trk_spec_layout_stamp(int64_t* out);

/* Fused FK + boolean collision fields: q [batch*horizon, D] -> in_collision [batch*horizon] (1 = at least one selected field
 * has a signed distance below its margin).  No link positions, costs or gradients are written.
 * reference: PlanningTask.compute_collision tasks.py:131-133 -> _compute_collision_or_cost(field_type='occupancy') tasks.py:139-232
 * = fk_map_collision (robot_base.py:171-174) + the OR of the fields' compute_embodiment_collision (distance_fields.py:210-215,
 * 283-291); get_trajs_collision_and_free calls it with margin=0. on the interpolated via points (tasks.py:247-251).
 * margin_override: NaN = each field's own margins (per-link margin + cutoff, per-pair margin), else this margin for every test.
 * link_pos_ws: DEVICE scratch [N, L, 3], used only when no generated kernel serves this model / cost model (then the
 * table-driven FK and field kernels run back to back); may be NULL otherwise. */
int trk_rollout_collision(const TrkModel* model, const TrkCostModel* cm, int32_t fields, const float* q, int64_t batch,
                          int32_t horizon, float margin_override, uint8_t* in_collision, float* link_pos_ws,
                          trk_stream_t stream);

/* element types of the reduced-precision entry points */
#define TRK_F32 0
#define TRK_F16 1

/* trk_rollout_cost_grad with q and link_pos_out stored as IEEE fp16 in HBM -- BASELINE config 5's "fp16 with fp32
 * cost accumulate": arithmetic, `cost` and `cost_block_sums` stay fp32; inputs are widened and outputs rounded once
 * (round-to-nearest-even) at the memory boundary.  Build-defined (the reference has no reduced-precision path); the
 * check is the fp64 oracle on the same fp16-rounded q.
 * The gradient: gq [B,H,D] of grad_dtype (TRK_F16, or TRK_F32 = the mixed mode: fp16 trajectories and positions, fp32
 * gradient) receives grad_scale * d cost / d q, the product formed in fp32; an fp16 store SATURATES at +-65504 -- it never
 * writes inf.  grad_scale (> 0, a power of two keeps the mantissa) is the caller's loss scale: the GP prior this gradient
 * is summed with (trk_gp_prior_cost_grad, same scale) reaches 1e5 .. 1e6 at config 5's sigma_gp = 0.1, dt = 5/128
 * (env_spheres_3d.py:57), beyond the fp16 range; the consumer divides by the scale (or folds it into its step size). */
int trk_rollout_cost_grad_f16(const TrkModel* model, const TrkCostModel* cm, const TrkRolloutWeights* w,
                              const void* q_f16, int64_t batch, int32_t horizon, void* link_pos_out_f16, float* cost,
                              void* gq, int32_t grad_dtype, float grad_scale, float* cost_block_sums, trk_stream_t stream);

/* BASELINE config 5's objective in ONE launch: trk_rollout_cost_grad(_f16) and the GP prior (trk_gp_prior_cost_grad) fused -- the
 * trajectory's q / qd rows are read once, gq and gqd are written once (no read-modify-write of the gradient between two launches).
 * BUILD-DEFINED like its two halves; checked against the fp64 oracle of both.
 *   q, qd [B,H,D] of io_dtype (TRK_F32 / TRK_F16; samples of a trajectory are consecutive, `horizon` of them);
 *   cost [B,H] (fp32) = the rollout's cost + weight/2 e_t^T Q^-1 e_t, the prior's factor between t and t+1 attributed to
 *       sample t (0 at t = H-1): sum_t cost[b,t] = the rollout's total + trk_gp_prior_cost_grad's cost[b];
 *   gq [B,H,D] = grad_scale (d rollout cost / d q + d prior / d q),  gqd [B,H,D] = grad_scale d prior / d qd, both of grad_dtype
 *       (fp16 stores saturate; fp32 trajectories: fp32 gradients, grad_scale = 1);  link_pos_out (nullable) [B,H,L,3] of io_dtype;
 *   cost_block_sums (nullable): as trk_rollout_cost_grad, of the combined cost.
 * A generated unit serves it with its subtrees scheduled one after the other (the dual Panda's arms: 128 registers, one resident
 * generation of wavefronts); where no unit does -- no unit for the robot, ring-staged robots, self pairs between the subtrees
 * with w_self != 0 -- the same result comes from the two-launch form (rollout, then the prior accumulated into its outputs). */
typedef struct TrkGpPrior {
    float dt, sigma, weight;
} TrkGpPrior;
int trk_rollout_gp_cost_grad(const TrkModel* model, const TrkCostModel* cm, const TrkRolloutWeights* w, const TrkGpPrior* gp,
                             const void* q, const void* qd, int64_t batch, int32_t horizon, int32_t io_dtype,
                             void* link_pos_out, float* cost, void* gq, void* gqd, int32_t grad_dtype, float grad_scale,
                             float* cost_block_sums, trk_stream_t stream);

/* trk_rollout_cost_grad with the collision fields evaluated on attached points instead of link origins: the cost
 * model's position columns (n_links_in, obj_link_idx, self_link_idx) index the points of `ps`; ee_link stays a LINK
 * index of the model.  point_pos_out [batch*horizon, n_points, 3] (nullable).  Same outputs otherwise. */
int trk_rollout_points_cost_grad(const TrkModel* model, const TrkPointSet* ps, const TrkCostModel* cm,
                                 const TrkRolloutWeights* w, const float* q, int64_t batch, int32_t horizon,
                                 float* point_pos_out, float* cost, float* gq, float* cost_block_sums,
                                 trk_stream_t stream);

/* reference: interpolate_traj_via_points trajectory/utils.py:37-50 (used by PlanningTask.get_trajs_collision_and_free
 * tasks.py:234-251): x [T, H, D] -> out [T, (H-1)*n_interp, D], out[t, i*n+a] = x[t,i]*alpha[a] + x[t,i+1]*beta[a];
 * alpha = linspace(0,1,n+2)[1:n+1], beta = 1 - alpha: DEVICE float[n_interp], computed by the caller. */
int trk_interpolate_via_points(const float* x, int64_t n_traj, int32_t horizon, int32_t dim, int32_t n_interp,
                               const float* alpha, const float* beta, float* out, trk_stream_t stream);

/* Trajectory validation in one pass over the via points: trk_interpolate_via_points + trk_rollout_collision without the
 * interpolated trajectories ever reaching HBM.  x [n_traj, horizon, state_dim] (the first D columns of a way point are the joint
 * positions, robot_base.py:148-149); sample (t, i, a) is x[t, i] * alpha[a] + x[t, i + 1] * beta[a], products and sum each rounded
 * once (trajectory/utils.py:47-49) -> in_collision [n_traj * (horizon - 1) * n_interp].
 * reference: PlanningTask.get_trajs_collision_and_free tasks.py:244-251 (margin_override = 0 there).
 * Served by the generated kernels only: TRK_ERR_UNSUPPORTED when no unit matches the model / cost model -- the caller then
 * runs the two-step form. */
int trk_rollout_collision_via(const TrkModel* model, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                              int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                              float margin_override, uint8_t* in_collision, trk_stream_t stream);

/* The same launch also folds the per-trajectory FLAGS of trk_traj_validate (round 6) -- bit 0 = some interpolated configuration of the
 * trajectory collides (tasks.py:255-256), bit 1 = some joint position x[t, h, d < D] lies outside [q_min[d], q_max[d]] (DEVICE float[D];
 * NaN = outside; tasks.py:270-273; every way point is an end of some segment the kernel has loaded anyway) -- as per-WAVEFRONT partial
 * results: partial_flags [trk_via_partial_flags_bytes(n_traj, horizon, n_interp)] uint8, a row of hi / 64 + 2 bytes per trajectory
 * (hi = (horizon - 1) n_interp samples), one byte per wavefront of 64 consecutive samples that holds samples of it; every byte a reader
 * looks at is written by plain stores on every call (no atomics, nothing to zero).  Hand the buffer to trk_traj_validate as `waypoint_collisions` with n_waypoints = -(horizon - 1) * n_interp: its partition
 * launch ORs the few bytes that cover a trajectory, and the separate flags launch disappears. */
int64_t trk_via_partial_flags_bytes(int64_t n_traj, int32_t horizon, int32_t n_interp);
int trk_rollout_collision_via_flags(const TrkModel* model, const TrkCostModel* cm, int32_t fields, const float* x, int64_t n_traj,
                                    int32_t horizon, int32_t state_dim, int32_t n_interp, const float* alpha, const float* beta,
                                    float margin_override, const float* q_min, const float* q_max, uint8_t* in_collision,
                                    uint8_t* partial_flags, trk_stream_t stream);

/* The rest of get_trajs_collision_and_free (tasks.py:253-299) on the device: three launches, no host round trip (two when
 * `waypoint_collisions` is the partial_flags buffer of trk_rollout_collision_via_flags and n_waypoints = -(samples per trajectory):
 * the partition launch assembles flags [n_traj] itself; q_min / q_max are then unused).
 *   flags [n_traj]      bit 0: some way-point byte of the trajectory is set (waypoint_collisions [n_traj, n_waypoints]);
 *                       bit 1: some joint position x[t, h, d < n_dofs] lies outside [q_min[d], q_max[d]] (NaN = outside)
 *   idx [n_traj rows]   a stable three-way partition of the trajectories, each group in increasing order:
 *                         rows [0, n_free)              flags == 0: collision free and inside the limits (tasks.py:255, 274, 282)
 *                         rows [n_free, +n_coll)        bit 0 set: colliding                             (tasks.py:256)
 *                         rows [.., +n_out)             flags == 2: collision free, outside the limits   (tasks.py:278-281)
 *                       int64 rows [t] (inner == 0) or [t / inner, t % inner] (a 4-D batch [n_traj / inner, inner, ...]),
 *                       like torch.argwhere.  The reference's `trajs_coll_idxs` is the last two groups together.
 *   counts [4]          DEVICE {n_free, n_coll, n_out, ticket}.  counts_host (nullable): PINNED HOST int32[4] that receives the
 *                       same numbers straight from the kernel (device-addressable under unified addressing), the ticket last
 *                       with system-scope release semantics: the host polls counts_host[3] for the `ticket` it passed -- no
 *                       copy call, no event -- and that is the only device -> host traffic of a validation.
 *   gathered (nullable) [n_traj, horizon, state_dim]: gathered[r] = x[idx[r]] (free trajectories first, then the others). */
int trk_traj_validate(const uint8_t* waypoint_collisions, const float* x, int64_t n_traj, int32_t horizon, int32_t state_dim,
                      int32_t n_waypoints, int32_t n_dofs, const float* q_min, const float* q_max, int64_t inner,
                      uint8_t* flags, int64_t* idx, int32_t* counts, int32_t* counts_host, int32_t ticket, float* gathered,
                      trk_stream_t stream);

/* reference: interpolate_points_v1 distance_fields.py:66-69 = F.interpolate(points^T, size=n_out, mode='linear',
 * align_corners=True)^T along the link axis (the link-sphere approximation of interpolate_link_pos, :145-147; also used by
 * robot_panda.py:199).  x [N, n_in, channels] -> out [N, n_out, channels],
 *     out[n, k, :] = w[2k] * x[n, src[2k], :] + w[2k+1] * x[n, src[2k+1], :];
 * src / w: DEVICE int32 / float [2 * n_out] -- ATen's source indices and fp32 weights, computed by the caller
 * (torch_robotics_amd/costmodel.py: interpolation_table).  _backward: gout [N, n_out, channels] -> gx [N, n_in, channels]. */
int trk_interpolate_columns(const float* x, int64_t n, int32_t n_in, int32_t channels, int32_t n_out, const int32_t* src,
                            const float* w, float* out, trk_stream_t stream);
int trk_interpolate_columns_backward(const float* gout, int64_t n, int32_t n_in, int32_t channels, int32_t n_out,
                                     const int32_t* src, const float* w, float* gx, trk_stream_t stream);

/* Constant-velocity Gaussian-process prior over trajectories -- the "GP-smoothness" term of BASELINE config 5.
 * BUILD-DEFINED, parity unpinned: the reference contains no such cost (its smoothness is finite differences,
 * trajectory/utils.py:53-64, metrics.py:27-35; it only carries the hyper-parameter names sigma_gp, dt for external
 * planners, env_spheres_3d.py:51-76).  The standard GPMP form is used:  per DOF x_t = (q_t, qd_t),
 * e_t = Phi(dt) x_t - x_{t+1},  Q^-1 = sigma^-2 [[12/dt^3, -6/dt^2], [-6/dt^2, 4/dt]],
 * cost[b] = weight * 1/2 sum_t sum_dof e_t^T Q^-1 e_t   and   gq, gqd = d cost / d q, d cost / d qd.
 * q, qd: [batch, horizon, dof] of `io_dtype` (TRK_F32 / TRK_F16; arithmetic and `cost` are fp32); gq, gqd: the same shape
 * of `grad_dtype` (TRK_F32, or TRK_F16 with fp16 trajectories) and receive grad_scale * the gradient (product in fp32; an
 * fp16 store saturates at +-65504, see trk_rollout_cost_grad_f16; `cost` is not scaled).  accumulate != 0 adds into
 * gq / gqd instead of overwriting -- to compose with trk_rollout_cost_grad(_f16)'s gradient, written with the same scale.
 * fp16-stored q puts a quantisation floor under this term: ulp(q)^2 / 12 * 12 / (sigma^2 dt^3) per
 * residual -- ~1.6 per time step and joint at sigma = 0.1, dt = 5/128, |q| in [1, 2). */
int trk_gp_prior_cost_grad(const void* q, const void* qd, int64_t batch, int32_t horizon, int32_t dof, int32_t io_dtype,
                           float dt, float sigma, float weight, float* cost, void* gq, void* gqd, int32_t grad_dtype,
                           float grad_scale, int32_t accumulate, trk_stream_t stream);

/* reference: finite_difference_vector trajectory/utils.py:53-64 (RobotBase.get_velocity / get_acceleration
 * robot_base.py:151-166): zero-padded differences along the horizon.  x, out [batch, horizon, dim];
 * method 0 = forward, 1 = backward, 2 = central. */
int trk_finite_difference(const float* x, int64_t batch, int32_t horizon, int32_t dim, float dt, int32_t method,
                          float* out, trk_stream_t stream);
/* reference: compute_path_length / compute_smoothness trajectory/metrics.py:7-12, 27-35:
 * out[b] = sum_t || x[b, t+1, c0:c0+dim] - x[b, t, c0:c0+dim] ||, x [batch, horizon, state_dim]. */
int trk_traj_diff_norm_sum(const float* x, int64_t batch, int32_t horizon, int32_t state_dim, int32_t c0, int32_t dim,
                           float* out, trk_stream_t stream);

/* Gauss-Newton normal equations from the geometric Jacobian of one link (the output of trk_fk_jacobian; reference:
 * DifferentiableTree.compute_forward_kinematics_and_geometric_jacobian robot_tree.py:218-248 stops at lin_jac / ang_jac).
 * BUILD-DEFINED: per sample J = [lin_jac; ang_jac] (6 x D):  JtJ [n, D, D] = J^T J,  Jtr [n, D] = J^T residual (nullable;
 * residual [n, 6] = [linear(3), angular(3)]) -- what a damped least-squares / Gauss-Newton IK step solves (JtJ + lambda I) dq = Jtr.
 * use_mfma != 0 computes JtJ with v_mfma_f32_4x4x1_16b_f32 (dof <= 8) instead of per-lane FMAs: the same values to fp32 rounding;
 * both variants exist so that the two can be measured side by side (the op is HBM-bound).
 * dq (nullable) [n, D]: the damped step itself, (JtJ + lambda I) dq = Jtr solved per sample by a Cholesky factorisation in the
 * kernel (needs `residual`); damping: DEVICE lambda, [n] with damping_stride 1 or one value with damping_stride 0 (NULL: 0). */
int trk_jtj(const float* lin_jac, const float* ang_jac, const float* residual, int64_t n, int32_t dof, int32_t use_mfma,
            float* JtJ, float* Jtr, const float* damping, int32_t damping_stride, float* dq, trk_stream_t stream);

/* The chain rule of the fused rollout under autograd: out[n, :] = g[n, :] * scale[n * scale_stride] -- the saved
 * d cost[n] / d q[n, :] of trk_rollout_cost_grad times the upstream gradient of cost[n] (what
 * `PlanningTask.compute_collision_cost(q).sum().backward()`, tasks.py:135-137, asks of the op's backward).  g / out [n, dim] of
 * io_dtype (TRK_F32 / TRK_F16); scale fp32: [n] with scale_stride 1, or ONE value with scale_stride 0 (`.sum().backward()`
 * hands down an expanded scalar: no per-row tensor is materialised for it); out may be g. */
int trk_scale_rows(const void* g, const float* scale, int32_t scale_stride, int64_t n, int32_t dim, int32_t io_dtype, void* out,
                   trk_stream_t stream);

/* Profiling hook (process-global, NULL = off): DEVICE uint64[ceil(N/64)][8]; the model-specialised fused kernel then
 * records the shader clock (s_memtime) of every wavefront at 8 phase boundaries (entry, q loaded, FK done, positions
 * staged, objects done, objectives done, reverse done, exit).  Used by tools/phase_profile.py; never set in production. */
int trk_debug_set_stamp_buffer(void* device_u64);

/* The one buffer a batch-sharded planner all-reduces: packed [1 + H + H*D] =
 *   [ sum of all costs | sum over trajectories of cost(b, h) | sum over trajectories of gq(b, h, d) ]
 * of ONE rank's evaluation -- cost [batch, horizon], gq [batch, horizon, dof] and cost_block_sums as written by
 * trk_rollout_cost_grad -- bit-reproducibly (fixed association order; packed[0] equals trk_reduce_sum of the block sums; two
 * small launches: partial column sums of row slices, then their fold).  gq is of grad_dtype (TRK_F32, or TRK_F16: the gradient of trk_rollout_cost_grad_f16) and holds grad_scale x the
 * gradient; the fp32 column sums are divided by grad_scale once, so `packed` is always unscaled fp32.  traj_cost (nullable)
 * [batch]: a per-trajectory cost (trk_gp_prior_cost_grad's) whose sum is added to packed[0].
 * scratch: DEVICE memory of trk_pack_sums_scratch_bytes(horizon, dof) bytes (the partial rows); one scratch per stream that may
 * run the call concurrently. */
int64_t trk_pack_sums_scratch_bytes(int32_t horizon, int32_t dof);
int trk_pack_sums(const float* cost, const void* gq, int32_t grad_dtype, float grad_scale, const float* cost_block_sums,
                  const float* traj_cost, int64_t batch, int32_t horizon, int32_t dof, float* scratch, float* packed,
                  trk_stream_t stream);

/* ---------------------------------------------------------------------------------
 * The exchange step of a batch-sharded planner as a PEER-TO-PEER MAILBOX ("alternative = peer-to-peer write of
 * 8 partials + local sum"; what is sharded is PlanningTask._compute_collision_or_cost's batch, tasks.py:206-230; the reference
 * itself is single-device and has no exchange).  One process per GPU; every rank owns a mailbox in device memory with
 * n_slots x world rows of n_floats; trk_mailbox_exchange stores the caller's packed row (trk_pack_sums' output) into the
 * rank's row of EVERY mailbox (one xGMI hop, all peers in parallel), raises a sequence flag, waits for the `world` flags of
 * its own mailbox and writes the sum of the `world` rows, added in rank order (bit-identical on every rank and run), to out.
 * trk_mailbox_exchange = trk_mailbox_send (the stores and the flag: never waits) + trk_mailbox_recv (the wait and the sum); a
 * planner sends right after the evaluation whose sums travel and receives some evaluations later, when the peers' rows have
 * arrived -- the wait leaves the critical path without a second stream.  Single-workgroup kernels, asynchronous on `stream`;
 * the sequence numbers live in device memory, so the calls can be captured into a hipGraph and replayed.  All ranks must issue
 * the same sequence of exchanges, with EQUAL (world, n_floats, n_slots) on every rank (the offsets of a peer's rows follow from
 * them: exchange the triple with the handles and compare before connect); on one rank send k, recv k, send k + 1, ... run in
 * that (stream) order, with bounded look-ahead: send k + a may be issued before recv k only for a <= (n_slots - 2) / 2 (two
 * slots = strict alternation, four = one send ahead) -- a send beyond that is refused (TRK_ERR_INVALID_ARG), as is a recv without
 * a send (the host counts launches; a captured graph counts once and must therefore be balanced).  A rank that waits longer than
 * TRK_MAILBOX_TIMEOUT_S (environment, default 5 s) for a flag gives up, counts a time-out (sticky, trk_mailbox_status) and writes
 * NaN to every element of `out` -- never an incomplete sum -- instead of hanging the GPU.
 *   create:     allocates the local mailbox (uncached, else fine-grained, else plain device memory -- the first kind that
 *               hipIpcGetMemHandle accepts; TRK_MAILBOX_ALLOC=uncached|finegrained|plain in the environment forces one).
 *   ipc_handle: writes the TRK_MAILBOX_HANDLE_BYTES bytes another process passes to connect (exchange them with any host
 *               transport, e.g. torch.distributed.all_gather_object).
 *   connect:    handles [world][TRK_MAILBOX_HANDLE_BYTES] (HOST; the entry of this rank is ignored): maps the peers' mailboxes.
 *   status:     synchronises with the device; exchanges issued, time-outs seen, allocation kind (0 uncached, 1 fine-grained, 2 plain).
 * world == 1 needs no connect.  2 <= n_slots <= 64, world <= 16, n_floats <= 2^22. */
#define TRK_MAILBOX_HANDLE_BYTES 64
typedef struct TrkMailbox TrkMailbox;
int trk_mailbox_create(int32_t world, int32_t rank, int32_t n_floats, int32_t n_slots, TrkMailbox** out);
int trk_mailbox_ipc_handle(const TrkMailbox* mb, void* handle /* host, TRK_MAILBOX_HANDLE_BYTES */);
int trk_mailbox_connect(TrkMailbox* mb, const void* handles /* host */);
int trk_mailbox_send(TrkMailbox* mb, const float* packed, trk_stream_t stream);
int trk_mailbox_recv(TrkMailbox* mb, float* out, trk_stream_t stream);
int trk_mailbox_exchange(TrkMailbox* mb, const float* packed, float* out, trk_stream_t stream);
int trk_mailbox_status(const TrkMailbox* mb, int64_t* n_exchanges, int64_t* n_timeouts, int32_t* alloc_kind);
void trk_mailbox_destroy(TrkMailbox* mb);

/* What a pointer VALUE is to this library, without dereferencing it: 0 = not a live handle (never created here, or destroyed),
 * 1 = TrkModel, 2 = TrkCostModel, 3 = TrkPointSet.  For bindings that carry handles as integers (a PyTorch dispatcher op's schema
 * has tensors and scalars only, csrc/trk_torch_ops.cpp): a stale integer becomes an error instead of a segfault.  Thread-safe. */
int trk_handle_kind(const void* handle);

/* Deterministic sum of n floats (fixed association order, one workgroup): x [n] -> out [1]. */
int trk_reduce_sum(const float* x, int64_t n, float* out, trk_stream_t stream);

/* ---------------------------------------------------------------------------------
 * SDF grid precompute (GridMapSDF.precompute_sdf grid_map_sdf.py:34-63): evaluates the analytic
 * objects of `cm` and their gradient at the linspace voxel centres.
 * -> sdf [nx,ny,nz], grad [nx,ny,nz,3] (device, caller-allocated). */
int trk_grid_precompute(const TrkCostModel* cm, const int32_t dims[3], const float lim_min[3],
                        const float lim_max[3], float* sdf, float* grad, trk_stream_t stream);

/* Signed distance of arbitrary points to the cost model's objects:
 * reference: ObjectField.compute_signed_distance primitives.py:387-405 (per object) -> sdf [N, n_objects]
 * points [N,3]. grad (nullable) [N, n_objects, 3]. */
int trk_sdf_points(const TrkCostModel* cm, const float* points, int64_t n, float* sdf, float* grad,
                   trk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRK_H */
