"""Drop-in for `torch_robotics.tasks.tasks.PlanningTask` on the hot path (tasks.py:22-232) plus the fused
entry point `rollout_cost_grad`, which runs FK + every configured objective + d cost / d q in ONE kernel."""
from __future__ import annotations

import os
import sys
from functools import partial

import numpy as np
import torch

from . import ops
from ._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
from .costmodel import CostModelSpec
from .environments import _np, objects_to_spec_parts, scene_version
from .fields import CollisionObjectDistanceField, CollisionWorkspaceBoundariesDistanceField


class Task:
    def __init__(self, env=None, robot=None, tensor_args=None, **kwargs):
        self.env, self.robot, self.tensor_args = env, robot, tensor_args


class PlanningTask(Task):
    def __init__(self, ws_limits=None, use_occupancy_map=False, cell_size=0.01, obstacle_cutoff_margin=0.01,
                 auto_specialize=True, clamp_sdf=False, interpolate_link_pos=None, **kwargs):
        super().__init__(**kwargs)
        # the first fused evaluation compiles + loads a generated kernel for this robot / collision model when none is
        # registered yet (~2 s with hipcc, cached on disk); TRK_NO_JIT=1 or auto_specialize=False keeps the table-driven path
        self.auto_specialize = bool(auto_specialize) and os.environ.get("TRK_NO_JIT", "0") != "1"
        self._jit_failed = False
        self.ws_limits = self.env.limits if ws_limits is None else ws_limits
        self.ws_min, self.ws_max = self.ws_limits[0], self.ws_limits[1]
        if use_occupancy_map:
            raise NotImplementedError        # tasks.py:160: the occupancy-map branch raises in the reference too
        self.use_occupancy_map = False
        self.obstacle_cutoff_margin = obstacle_cutoff_margin
        r = self.robot
        self.df_collision_self = r.df_collision_self
        # clamp_sdf (not a keyword of the reference's PlanningTask, which builds its fields with the default False): the three
        # fields become hinges relu(margin - sdf), the form an optimiser needs -- the plain cost decreases without bound
        self.clamp_sdf = bool(clamp_sdf)
        # interpolate_link_pos (extension keyword; None = automatic): a robot declared with more collision points than links
        # (robot_base.py:57-73 repeats its margins accordingly) needs fields that interpolate, distance_fields.py:145-147
        if interpolate_link_pos is None:
            interpolate_link_pos = (r.num_interpolated_points_for_object_collision_checking !=
                                    len(r.link_idxs_for_object_collision_checking))
        self.interpolate_link_pos = bool(interpolate_link_pos)
        common = dict(link_idxs_for_collision_checking=r.link_idxs_for_object_collision_checking, clamp_sdf=self.clamp_sdf,
                      interpolate_link_pos=self.interpolate_link_pos,
                      num_interpolated_points=r.num_interpolated_points_for_object_collision_checking,
                      link_margins_for_object_collision_checking_tensor=r.link_margins_for_object_collision_checking_tensor,
                      cutoff_margin=obstacle_cutoff_margin, tensor_args=self.tensor_args)
        self.df_collision_objects = CollisionObjectDistanceField(r, df_obj_list_fn=self.env.get_df_obj_list, **common)
        if self.env.obj_extra_list is not None:
            self.df_collision_extra_objects = CollisionObjectDistanceField(
                r, df_obj_list_fn=partial(self.env.get_df_obj_list, return_extra_objects_only=True), **common)
            self._collision_fields_extra_objects = [self.df_collision_extra_objects]
        else:
            self._collision_fields_extra_objects = []
        self.df_collision_ws_boundaries = CollisionWorkspaceBoundariesDistanceField(
            r, ws_bounds_min=self.ws_min, ws_bounds_max=self.ws_max, **common)
        self._collision_fields = [self.df_collision_self, self.df_collision_objects, self.df_collision_ws_boundaries]
        self._fused = None
        self._ee = dict(target=None, w_pos=1.0, w_rot=1.0, square=True, link=None)

    def get_collision_fields(self):
        return self._collision_fields

    def get_collision_fields_extra_objects(self):
        return self._collision_fields_extra_objects

    def distance_q(self, q1, q2):
        return self.robot.distance_q(q1, q2)

    def sample_q(self, without_collision=True, **kwargs):      # tasks.py:97-101
        if without_collision:
            return self.random_coll_free_q(**kwargs)
        return self.robot.random_q(**kwargs)

    def random_coll_free_q(self, n_samples=1, max_samples=1000, max_tries=1000):   # tasks.py:103-129
        """`n_samples` collision-free configurations by rejection: rounds of `max_samples` uniform draws through the fused FK +
        boolean-field kernel, the survivors of a round compacted with one masked gather.  The draws are i.i.d., so the first
        survivors of a round are as good as the reference's random subset of them.  Returns (n_samples, q_dim).squeeze() like
        the reference, and ends the process like it when `max_tries` rounds do not suffice."""
        kept, missing = [], int(n_samples)
        for _ in range(max_tries):
            qs = self.robot.random_q(max_samples)
            free = qs[~self.compute_collision(qs).reshape(-1)][:missing]
            if free.shape[0]:
                kept.append(free)
                missing -= int(free.shape[0])
            if missing <= 0:
                return torch.cat(kept, dim=0).to(torch.float32).squeeze()
        sys.exit("Could not find a collision free configuration")

    # ---------------------------------------------------------------------------------------------
    # one cost model for the fused kernel and for compute_collision(_cost)
    # ---------------------------------------------------------------------------------------------
    def set_ee_target(self, target_H, w_pos=1.0, w_rot=1.0, square=True, link_name=None):
        """End-effector SE(3) tracking term of the fused rollout (EESE3DistanceField semantics)."""
        tree = self.robot.diff_panda
        link = tree._name_to_idx_map[link_name or self.robot.link_name_ee]
        new = dict(target=_np(target_H).astype(np.float32).reshape(4, 4), w_pos=w_pos, w_rot=w_rot, square=square, link=link)
        same_cfg = all(self._ee[k] == new[k] for k in ("w_pos", "w_rot", "square", "link"))
        self._ee = new
        if self._fused is not None and same_cfg:
            self._fused[1].set_ee_target(new["target"])
        else:
            self._fused = None

    @property
    def _has_tree(self) -> bool:
        """False for robots without kinematics (RobotPointMass3D: task space == configuration space)."""
        return getattr(self.robot, "diff_panda", None) is not None

    def _n_columns(self) -> int:
        """Width of fk_map_collision's output: the links, plus the grasped object's points if there is one."""
        if not self._has_tree:
            if self.robot.q_dim != 3:
                raise NotImplementedError("2-D point robots are outside the 3-D hot path")
            return 1
        if getattr(self.robot, "has_extra_points", False):
            return len(self.robot.collision_point_set()[0])
        return self.robot.diff_panda._kin.n_links

    def build_cost_spec(self) -> CostModelSpec:
        r = self.robot
        spec = CostModelSpec(n_links_in=self._n_columns())
        spec.obj_link_idx = self.df_collision_objects._columns(spec.n_links_in, spec)
        spec.obj_link_margin = self.df_collision_objects._margin_vector(len(spec.obj_link_idx))
        spec.objects, spec.grid = objects_to_spec_parts(self.env.get_df_obj_list())
        spec.ws_min, spec.ws_max = _np(self.ws_min).astype(np.float32), _np(self.ws_max).astype(np.float32)
        if self.df_collision_self is not None:
            self.df_collision_self._fill_spec(spec)
        for fld, bit in ((self.df_collision_self, FIELD_SELF), (self.df_collision_objects, FIELD_OBJECTS),
                         (self.df_collision_ws_boundaries, FIELD_WS)):
            if fld is not None and (self.clamp_sdf or getattr(fld, "clamp_sdf", False)):
                spec.clamp_fields |= bit
        if self._ee["target"] is not None:
            spec.ee_link, spec.ee_target = self._ee["link"], self._ee["target"]
            spec.ee_w_pos, spec.ee_w_rot, spec.ee_square = self._ee["w_pos"], self._ee["w_rot"], self._ee["square"]
        return spec

    def _fused_handles(self, device):
        # keyed by device AND the scene's current object poses: the reference reads each object's pose on every evaluation,
        # so `obj.set_position_orientation(...)` between two calls must be seen here too (a moved object rebuilds the tables)
        key = (str(device), scene_version(self.env.get_df_obj_list()))
        if self._fused is None or self._fused[2] != key:
            spec = self.build_cost_spec()
            if not self._has_tree:                         # no kinematics: only the cost model is needed
                self._fused = (None, ops.CostHandle(spec, device), key)
                return self._fused[0], self._fused[1]
            if self._jit_failed and getattr(self, "_jit_error", None) is not None and os.environ.get("TRK_ALLOW_TABLE_DRIVEN", "0") != "1":
                raise RuntimeError(f"PlanningTask: no generated kernel serves this robot / collision model and compiling one failed earlier "
                                   f"({type(self._jit_error).__name__}: {self._jit_error}).  Set TRK_ALLOW_TABLE_DRIVEN=1 to run the ~10 x "
                                   f"slower table-driven kernels instead.") from self._jit_error
            self._fused = (self.robot.diff_panda._handle, ops.CostHandle(spec, device), key)
            from . import jit
            if self.auto_specialize and not self._jit_failed and jit.generatable(spec, getattr(self.robot, "has_extra_points", False)):
                try:                                   # a unit whose template equals this cost model may already exist
                    kin = self.robot.diff_panda._kin
                    if getattr(self.robot, "has_extra_points", False):
                        pl, po = self.robot.collision_point_set()
                        if not jit.has_matching_points_unit(kin, pl, po, spec):
                            jit.specialize_points(kin, pl, po, spec)
                    elif not jit.has_matching_unit(kin, spec):
                        jit.specialize_for_cost_spec(kin, spec)
                except Exception as e:
                    # No compiler / a compile error.  The table-driven kernels compute the same function ~10 x slower (7 % of the
                    # HBM roofline), so running them silently is a performance bug: it is an error unless the caller allows it.
                    self._jit_failed = True
                    if os.environ.get("TRK_ALLOW_TABLE_DRIVEN", "0") != "1":
                        self._fused = None          # EVERY later call raises too (below): a caught / retried error must not degrade silently
                        self._jit_error = e
                        raise RuntimeError(
                            f"PlanningTask: no generated kernel serves this robot / collision model and compiling one failed "
                            f"({type(e).__name__}: {e}).  Set TRK_ALLOW_TABLE_DRIVEN=1 (or auto_specialize=False / TRK_NO_JIT=1) to run "
                            f"the ~10 x slower table-driven kernels instead.") from e
                    import warnings
                    warnings.warn(f"run-time kernel specialisation failed ({type(e).__name__}: {e}); TRK_ALLOW_TABLE_DRIVEN=1: "
                                  f"using the table-driven kernels")
        return self._fused[0], self._fused[1]

    def specialize(self, verbose: bool = False):
        """Compile (once, cached on disk) and load a generated fused kernel for THIS robot and collision model
        (torch_robotics_amd/jit.py).  The robots of the benchmark configs ship with one; any other URDF gets the
        table-driven kernels until this is called.  Returns the unit identifier, or None when the collision columns are
        attached points (those have their own ahead-of-time units)."""
        from . import jit
        if getattr(self.robot, "has_extra_points", False):
            return None
        ident = jit.specialize_for_cost_spec(self.robot.diff_panda._kin, self.build_cost_spec(), verbose)
        return ident

    def _points(self, device):
        """PointSetHandle when the collision columns are not simply the links (grasped object), else None."""
        if not getattr(self.robot, "has_extra_points", False):
            return None
        return self.robot._point_set(device)

    @ops.host_round_trip
    def rollout_cost_grad(self, x, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=True, cost_sum=None, out=None):
        """Fused FK + objectives + gradient.  x (B,H,>=D) or (N,>=D) -> (link_pos, cost, d cost/d q)."""
        q = self.robot.get_position(x)
        model, cm = self._fused_handles(q.device)
        ps = self._points(q.device)
        if ps is not None:
            if out is not None:
                raise NotImplementedError("pre-allocated outputs are only supported without a grasped object")
            return ops.rollout_points_cost_grad(ps, cm, (w_self, w_obj, w_ws, w_ee), q, want_pos=want_pos, cost_sum=cost_sum)
        return ops.rollout_cost_grad(model, cm, (w_self, w_obj, w_ws, w_ee), q, want_pos=want_pos, cost_sum=cost_sum, out=out)

    def rollout_plan(self, q, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=True) -> "ops.RolloutPlan":
        """Pre-bound fused evaluation for a planner's inner loop: buffers and arguments are resolved once, `plan.launch()`
        is one C call (~3 us of host time instead of ~18 us through `rollout_cost_grad`); results land in
        `plan.link_pos / plan.cost / plan.gq`.  q (B,H,D) is read in place on every launch (update it between launches)."""
        model, cm = self._fused_handles(q.device)
        strict = os.environ.get("TRK_ALLOW_TABLE_DRIVEN", "0") != "1"
        ps = self._points(q.device)
        if ps is not None:          # grasped object / link spheres: the columns are attached points (round 6: pre-bound like the link models)
            return ops.PointsRolloutPlan(ps, cm, (w_self, w_obj, w_ws, w_ee), q, want_pos=want_pos, strict=strict)
        return ops.RolloutPlan(model, cm, (w_self, w_obj, w_ws, w_ee), q, want_pos=want_pos, strict=strict)

    def rollout_gp_plan(self, q, qd, dt, sigma_gp, gp_weight=1.0, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False,
                        grad_dtype=None, grad_scale=1.0) -> "ops.RolloutGpPlan":
        """The planner's whole objective as ONE pre-bound launch (`trk_rollout_gp_cost_grad`): the collision / EE terms of
        `rollout_plan` plus the constant-velocity GP prior on (q, qd) -- `plan.cost` (B,H) incl. the prior's factor costs, `plan.gq`,
        `plan.gqd`.  q, qd (B,H,D) fp32 or fp16 are read in place on every launch."""
        if self._points(q.device) is not None:
            raise NotImplementedError("rollout_gp_plan is for link-column cost models (no grasped object / link spheres)")
        model, cm = self._fused_handles(q.device)
        return ops.RolloutGpPlan(model, cm, (w_self, w_obj, w_ws, w_ee), q, qd, dt, sigma_gp, gp_weight, want_pos=want_pos,
                                 grad_dtype=grad_dtype, grad_scale=grad_scale, strict=os.environ.get("TRK_ALLOW_TABLE_DRIVEN", "0") != "1")

    def capture_cost_backward(self, x, reduce=torch.sum, warmup: int = 3) -> "GraphedCostBackward":
        """`reduce(task.compute_collision_cost(x)).backward()` captured ONCE as a hipGraph (a planner's inner loop calls it with
        the same shapes thousands of times; eagerly the autograd engine's hand-offs cost ~10x the kernels).
        x must be a leaf tensor that requires grad and keeps its storage: update it IN PLACE between replays (an optimiser's
        `step()` does).  `replay()` leaves the gradient in `x.grad` and returns the per-sample cost tensor of the capture."""
        return GraphedCostBackward(self, x, reduce, warmup)

    # ---------------------------------------------------------------------------------------------
    @ops.host_round_trip
    def compute_collision(self, x, **kwargs):                  # tasks.py:131-133
        return self._compute_collision_or_cost(self.robot.get_position(x), field_type="occupancy", **kwargs)

    @ops.host_round_trip
    def compute_collision_cost(self, x, **kwargs):             # tasks.py:135-137
        return self._compute_collision_or_cost(self.robot.get_position(x), field_type="sdf", **kwargs)

    def _compute_collision_or_cost(self, q, field_type="occupancy", **kwargs):   # tasks.py:139-232
        if q.ndim == 1:
            q = q.unsqueeze(0).unsqueeze(0)
        elif q.ndim == 2:
            q = q.unsqueeze(1)
        elif q.ndim > 3:
            raise NotImplementedError
        model, cm = self._fused_handles(q.device)
        fields = FIELD_OBJECTS | FIELD_WS | (FIELD_SELF if self.df_collision_self is not None else 0)
        if model is None:                                  # RobotPointMass3D: positions are q itself (robot_point_mass.py:29-32)
            pos = self.robot.fk_map_collision(q)
            if field_type == "occupancy":
                return ops.collision_fields(cm, fields, pos.detach(), margin=kwargs.get("margin", None)).reshape(q.shape[:-1])
            return ops.cost_fields_ad(cm, fields, pos).reshape(q.shape[:-1])
        ps = self._points(q.device)
        if field_type == "occupancy":
            if ps is None:                                    # FK + boolean fields in one launch, one byte per sample out
                return ops.rollout_collision(model, cm, fields, q.detach(), margin=kwargs.get("margin", None))
            pos = ops.fk_points(ps, q.detach())
            return ops.collision_fields(cm, fields, pos, margin=kwargs.get("margin", None)).reshape(q.shape[:-1])
        w = (1.0 if self.df_collision_self is not None else 0.0, 1.0, 1.0, 0.0)
        if (torch.is_grad_enabled() and q.requires_grad) or ops._dispatch():
            # one fused kernel; backward reuses its gradient.  Under a compiler (torch.compile) also without autograd: the dispatcher
            # op is what a tracer can see (the ctypes call below is not)
            cost, _ = ops.rollout_ad(model, cm, w, q, ps, want_pos=False)
            return cost
        if ps is not None:
            return ops.rollout_points_cost_grad(ps, cm, w, q, want_pos=False)[1]
        _, cost, _ = ops.rollout_cost_grad(model, cm, w, q, want_pos=False)
        return cost

    # ---------------------------------------------------------------------------------------------
    # trajectory validation (tasks.py:234-328): everything up to the final slicing runs on the device -- via-point
    # interpolation fused into the FK + boolean-field kernel, then per-trajectory flags, the ordered index lists and the two
    # gathers (ops.traj_validate) -- and the host reads three counters once.  Return values and shapes are the reference's.
    # ---------------------------------------------------------------------------------------------
    def _waypoint_collisions(self, flat, num_interpolation):
        """bool (T, W) for trajectories flat (T, H, S): margin 0 on the interpolated via points (tasks.py:244-251)."""
        return self._waypoint_collisions_and_flags(flat, num_interpolation)

    def _waypoint_collisions_and_flags(self, flat, num_interpolation, limits=None):
        """`_waypoint_collisions`; with limits=(q_min, q_max): -> (bool (T, W), flags or None) -- the per-trajectory flags folded into
        the same launch when a generated kernel serves the call (ops.rollout_collision_via)."""
        if self._has_tree and self._points(flat.device) is None and num_interpolation > 0 and flat.shape[1] >= 2:
            model, cm = self._fused_handles(flat.device)
            fields = FIELD_OBJECTS | FIELD_WS | (FIELD_SELF if self.df_collision_self is not None else 0)
            res = ops.rollout_collision_via(model, cm, fields, flat, num_interpolation, margin=0.,
                                            limits=limits if (limits is not None and limits[0].numel() == model.n_dofs) else None)
            if res is not None:
                if limits is None:
                    return res
                return res if isinstance(res, tuple) else (res, None)
        wp = self.compute_collision(ops.interpolate_traj_via_points(flat, num_interpolation=num_interpolation), margin=0.)
        return wp if limits is None else (wp, None)

    @ops.host_round_trip
    def get_trajs_collision_and_free(self, trajs, return_indices=False, num_interpolation=5):
        assert trajs.ndim == 3 or trajs.ndim == 4
        batched = trajs.ndim == 4                       # (goals or steps, batch, horizon, state)
        lead = tuple(trajs.shape[:-2])
        H, S = int(trajs.shape[-2]), int(trajs.shape[-1])
        flat = trajs.detach().reshape(-1, H, S)
        if flat.dtype != torch.float32 or not flat.is_contiguous():
            flat = flat.to(torch.float32).contiguous()
        lim = getattr(self, "_q_lim", None)             # the limits as fp32 device vectors, keyed by the tensors they came from
        if lim is None or lim[0] is not self.robot.q_min or lim[1] is not self.robot.q_max or lim[2].device != flat.device:
            lim = self._q_lim = (self.robot.q_min, self.robot.q_max,
                                 self.robot.q_min.to(flat.device, torch.float32).contiguous(),
                                 self.robot.q_max.to(flat.device, torch.float32).contiguous())
        # round 6: the via-point launch also produces the per-trajectory flags (collision, joint limits) -- one launch less
        if type(self)._waypoint_collisions is PlanningTask._waypoint_collisions:
            wp, flags = self._waypoint_collisions_and_flags(flat, num_interpolation, limits=(lim[2], lim[3]))
        else:                                           # a subclass supplies the way-point collisions itself: the three-launch form
            wp, flags = self._waypoint_collisions(flat, num_interpolation), None
        part = ops.traj_validate(wp, flat, self.robot.q_dim, lim[2], lim[3], inner=lead[1] if batched else 0, flags=flags)
        n_free, n_coll, n_out = part.counts()                                   # the one host synchronisation
        # the partition is [free | colliding | collision free but outside the limits]; the reference's second list is
        # "colliding, then the limit violators" -- except that it is only the violators when no trajectory is free although
        # some were collision free (tasks.py:275-276 replaces the list), and the limits are not looked at when none was (:264)
        lo, hi = n_free, n_free + n_coll + n_out
        if n_free == 0 and n_out:
            lo = n_free + n_coll
        free_idxs, coll_idxs = part.idx[:n_free], part.idx[lo:hi]
        if n_free == 1 and not batched:                  # argwhere(...).squeeze() of one hit indexes out a 1-D row (tasks.py:274)
            free_idxs = free_idxs.reshape(-1)
        trajs_free = part.gathered[:n_free] if n_free else None
        trajs_coll = part.gathered[lo:hi] if hi > lo else None
        if return_indices:
            return trajs_coll, coll_idxs, trajs_free, free_idxs, wp.reshape(lead + (wp.shape[-1],))
        return trajs_coll, trajs_free

    def compute_fraction_free_trajs(self, trajs, **kwargs):
        _, coll_idxs, _, free_idxs, _ = self.get_trajs_collision_and_free(trajs, return_indices=True)
        n_free, n_coll = free_idxs.nelement(), coll_idxs.nelement()
        return n_free / (n_free + n_coll)

    def compute_collision_intensity_trajs(self, trajs, **kwargs):
        _, _, _, _, wp = self.get_trajs_collision_and_free(trajs, return_indices=True)
        return torch.count_nonzero(wp) / wp.nelement()

    def compute_success_free_trajs(self, trajs, **kwargs):
        _, trajs_free = self.get_trajs_collision_and_free(trajs)
        return 1 if (trajs_free is not None and trajs_free.nelement() >= 1) else 0


class GraphedCostBackward:
    """One planner iteration's cost + gradient, replayable (PlanningTask.capture_cost_backward).

    `reduce=torch.sum` (the idiom of tasks.py:135-137 under autograd, `compute_collision_cost(x).sum().backward()`) is RECOGNISED: the
    gradient of the sum of the costs IS the `gq` the fused rollout writes, so a replay is ONE launch of that kernel with `x.grad`'s
    storage as its gradient output -- no `sum`, no `ones`, no `scale_rows`, no graph (a pre-bound C-ABI call costs the host less than a
    one-node graph launch); bit-identical to the eager idiom, whose backward multiplies the same `gq` by 1.0.  (Round 4 replayed a
    captured rollout + sum + expand + scale_rows: 29.8 us for a 9.3 us kernel.)

    Any other reduction follows torch's recipe for whole-network capture: a few eager iterations on a side stream (allocator and
    autograd warm-up, first-use kernel loads), `x.grad` reset to None so that the backward inside the capture allocates it from the
    graph's private pool, then `torch.cuda.graph`.  Afterwards `x.grad` and `cost` are static tensors a replay refills."""

    def __init__(self, task: "PlanningTask", x: torch.Tensor, reduce, warmup: int):
        if not (x.is_leaf and x.requires_grad and x.is_cuda):
            raise ValueError("capture_cost_backward: x must be a CUDA leaf tensor with requires_grad=True")
        self.x = x
        self.plan = self.graph = self._total = None
        D = getattr(task.robot, "q_dim", None)
        if (reduce is torch.sum and task._has_tree and task._points(x.device) is None and x.dtype == torch.float32 and x.is_contiguous()
                and x.dim() in (2, 3) and x.shape[-1] == D and x.numel() > 0 and os.environ.get("TRK_NO_SUM_FASTPATH", "0") != "1"):
            model, cm = task._fused_handles(x.device)
            q3 = x.detach() if x.dim() == 3 else x.detach().unsqueeze(1)
            w = (1.0 if task.df_collision_self is not None else 0.0, 1.0, 1.0, 0.0)      # compute_collision_cost's weights
            x.grad = torch.empty_like(x)
            self.plan = ops.RolloutPlan(model, cm, w, q3, want_pos=False, gq_out=x.grad, strict=False)    # the eager idiom's own dispatch rules apply
            assert self.plan.q.data_ptr() == x.data_ptr()        # the plan reads x's storage in place
            self.block_sums = torch.zeros(ops.n_blocks(q3.shape[0] * q3.shape[1]), device=x.device, dtype=torch.float32)
            self._bs_ptr = self.block_sums.data_ptr()
            self.cost, self.grad = self.plan.cost, x.grad
            self.plan.launch(self._bs_ptr)
            return
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                x.grad = None
                reduce(task.compute_collision_cost(x)).backward()
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        x.grad = None
        with torch.cuda.graph(self.graph):
            self.cost = task.compute_collision_cost(x)
            self._total = reduce(self.cost)
            self._total.backward()
        self.grad = x.grad

    @property
    def total(self) -> torch.Tensor:
        """reduce(cost) of the latest replay (sum fast path: evaluated on request with torch.sum -- the eager idiom's bits --; the
        kernel's own per-wavefront partial sums are in `block_sums`)."""
        return self.cost.sum() if self.plan is not None else self._total

    def replay(self) -> torch.Tensor:
        """Re-evaluate at the current contents of x: fills `x.grad` (== `self.grad`) and returns `self.cost`."""
        if self.plan is not None:
            if self.x.grad is not self.grad:                 # the caller dropped / replaced x.grad (zero_grad(set_to_none=True)): rebind it
                self.x.grad = self.grad
            self.plan.launch(self._bs_ptr)
            return self.cost
        self.graph.replay()
        return self.cost
