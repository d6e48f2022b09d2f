#!/usr/bin/env python3
"""The user-facing calls at 4096 x 64 (host + GPU time per call): PlanningTask.compute_collision_cost with and without autograd,
compute_collision, RobotPanda.fk_map_collision / get_EE_pose."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
qg = q.clone().requires_grad_(True)
def t(name, fn, n=300):
    """median of five blocks of n calls after 100 warm-up calls (the first milliseconds of a process -- clocks, the autograd engine's
    device thread, allocator pools -- measured up to 25 us per call slower than the steady state)"""
    for _ in range(100): fn()
    blocks = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / n * 1e6)
    blocks.sort()
    print(f"{name:58s} {blocks[2]:8.1f} us per call   (blocks {blocks[0]:.1f} .. {blocks[-1]:.1f})")
t("task.compute_collision_cost(q)            [no grad]", lambda: task.compute_collision_cost(q))
def fb():
    qg.grad = None
    task.compute_collision_cost(qg).sum().backward()
t("task.compute_collision_cost(q).sum().backward()", fb)
def floor():
    qg.grad = None
    (qg * 2.0).sum().backward()
t("[torch floor] (q * 2).sum().backward()  -- no op of ours", floor)
torch.autograd.set_multithreading_enabled(False)
t("task.compute_collision_cost(q).sum().backward(), autograd multithreading off", fb)
t("[torch floor] (q * 2).sum().backward(), autograd multithreading off", floor)
torch.autograd.set_multithreading_enabled(True)
t("task.compute_collision(q)", lambda: task.compute_collision(q))
t("robot.fk_map_collision(q)", lambda: robot.fk_map_collision(q))
t("robot.get_EE_pose(q)", lambda: robot.get_EE_pose(q.reshape(-1, 7)))
t("tree.compute_forward_kinematics_all_links(q)", lambda: robot.diff_panda.compute_forward_kinematics_all_links(q.reshape(-1, 7)))
t("task.get_trajs_collision_and_free(4096 x 64, 5 via points)", lambda: task.get_trajs_collision_and_free(q, num_interpolation=5), n=100)
from torch_robotics_amd import ops
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
m_, cm_ = task._fused_handles(dev)
t("  ops.rollout_collision_via (fused interpolation + FK + boolean fields)", lambda: ops.rollout_collision_via(m_, cm_, FIELD_OBJECTS | FIELD_WS | FIELD_SELF, q, 5, margin=0.0), n=100)
wp_ = ops.rollout_collision_via(m_, cm_, FIELD_OBJECTS | FIELD_WS | FIELD_SELF, q, 5, margin=0.0)
qmin_, qmax_ = robot.q_min.to(dev).contiguous(), robot.q_max.to(dev).contiguous()
t("  ops.traj_validate, no host read (flags + partition + gathers)", lambda: ops.traj_validate(wp_, q, 7, qmin_, qmax_), n=100)
t("  ops.traj_validate + counts() (the one host read)", lambda: ops.traj_validate(wp_, q, 7, qmin_, qmax_).counts(), n=100)
# round 6: the per-trajectory flags folded into the via-point launch (what get_trajs_collision_and_free runs now)
t("  ops.rollout_collision_via(limits=...) (+ per-wavefront partial flags)", lambda: ops.rollout_collision_via(m_, cm_, FIELD_OBJECTS | FIELD_WS | FIELD_SELF, q, 5, margin=0.0, limits=(qmin_, qmax_)), n=100)
_, fl_ = ops.rollout_collision_via(m_, cm_, FIELD_OBJECTS | FIELD_WS | FIELD_SELF, q, 5, margin=0.0, limits=(qmin_, qmax_))
t("  ops.traj_validate(flags=...), no host read (partition + gathers)", lambda: ops.traj_validate(None, q, 7, qmin_, qmax_, flags=fl_), n=100)
t("  ops.traj_validate(flags=...) + counts()", lambda: ops.traj_validate(None, q, 7, qmin_, qmax_, flags=fl_).counts(), n=100)

# the same idiom through the dispatcher ops (torch.ops.trk.*; what torch.compile sees) and as a captured hipGraph
ops._ALWAYS_DISPATCH = True
t("[dispatcher ops] task.compute_collision_cost(q)   [no grad]", lambda: task.compute_collision_cost(q))
t("[dispatcher ops] task.compute_collision_cost(q).sum().backward()", fb)
ops._ALWAYS_DISPATCH = False
try:
    cfn = torch.compile(task.compute_collision_cost, fullgraph=True)       # one graph: the native op takes the handles as integers
    cfn(q); cfn(qg).sum().backward()
    t("[torch.compile] compute_collision_cost(q)   [no grad]", lambda: cfn(q))
    def fbc():
        qg.grad = None
        cfn(qg).sum().backward()
    t("[torch.compile] compute_collision_cost(q).sum().backward()", fbc)
    cfloor = torch.compile(lambda x: x * 2.0, fullgraph=True)             # what a compiled function + its compiled backward cost with no op of ours
    cfloor(qg).sum().backward()
    def fbf():
        qg.grad = None
        cfloor(qg).sum().backward()
    t("[torch floor, compiled] torch.compile(lambda x: x * 2)(q).sum().backward()", fbf)
except Exception as e:
    print("torch.compile path failed:", type(e).__name__, str(e)[:200])
# whole-iteration hipGraph: forward + backward of the idiom captured once, replayed per planner iteration
qs = q.clone().requires_grad_(True)
graphed = task.capture_cost_backward(qs)
t("[hipGraph replay] task.capture_cost_backward(q).replay()  (= compute_collision_cost(q).sum().backward())", graphed.replay)
plan = task.rollout_plan(q, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False)
t("[reference] pre-bound fused kernel, cost + gradient (RolloutPlan.launch)", plan.launch, n=1000)
