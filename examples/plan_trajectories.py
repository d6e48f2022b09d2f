#!/usr/bin/env python3
"""A small batch trajectory optimiser on the fused kernels -- how an external planner (the consumers of the reference's
`PlanningTask.get_collision_fields()` / `compute_collision_cost`) sits on this package.

B straight-line initial trajectories from a start configuration to B sampled collision-free goals are improved by Adam
on   w_obj * (self + object + workspace collision hinges)  +  (constant-velocity GP prior on q, qd)
where every cost / gradient evaluation is ONE launch of `trk_rollout_gp_cost_grad` over all (B x H) configurations (the fused rollout
with the GP prior fused in: `task.rollout_gp_plan`); the result is validated the way the reference does it after planning
(`get_trajs_collision_and_free`: 5 via points per segment, fused FK + boolean fields).  Needs the MI355X: there is no CPU path.

    python examples/plan_trajectories.py [--batch 256] [--horizon 64] [--iters 200]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch

import torch_robotics_amd as tra
from torch_robotics_amd import ops


def main(batch=256, horizon=64, iters=200, device="cuda:0", verbose=True, seed=0):
    torch.manual_seed(seed)
    ta = dict(device=torch.device(device), dtype=torch.float32)
    robot = tra.RobotPanda(tensor_args=ta)
    # clamp_sdf=True: the fields are hinges relu(margin - sdf) (distance_fields.py:114-117) -- what an optimiser needs; the plain
    # margin - sdf that the reference's PlanningTask sums decreases without bound away from the obstacles
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.05, clamp_sdf=True,
                            tensor_args=ta)
    T, dt = 5.0, 5.0 / horizon
    q_start = task.random_coll_free_q(n_samples=1).reshape(1, 1, -1)
    q_goal = task.random_coll_free_q(n_samples=batch).reshape(batch, 1, -1)
    s = torch.linspace(0.0, 1.0, horizon, **ta).reshape(1, horizon, 1)
    q = (q_start + s * (q_goal - q_start)).contiguous()                  # (B, H, D) straight lines in configuration space
    qd = ((q_goal - q_start) / T).expand(batch, horizon, -1).contiguous()

    w_obj, sigma_gp, lr = 50.0, 2.0, 1e-2
    # pre-bound launch: reads q, qd in place; cost = w_obj * (self + object + workspace hinges) + the prior's factor costs
    plan = task.rollout_gp_plan(q, qd, dt, sigma_gp, gp_weight=1.0, w_self=w_obj, w_obj=w_obj, w_ws=w_obj)
    free_mask = torch.ones(1, horizon, 1, **ta)
    free_mask[:, 0] = 0.0
    free_mask[:, -1] = 0.0                                               # start and goal stay fixed
    opt = torch.optim.Adam([q, qd], lr=lr)                               # the GP prior is stiff (1/dt^3): plain descent diverges
    hist = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        plan.launch()                                                    # plan.cost (B,H), plan.gq, plan.gqd (B,H,D)
        if verbose and (it % 50 == 0 or it == iters - 1):
            # the prior's share of plan.cost, from the SAME q / qd the launch read (before the step moves them in place)
            c_gp = ops.gp_prior_cost_grad(q, qd, dt, sigma_gp)[0]
            hist.append((it, float((plan.cost.sum(1) - c_gp).mean()) / w_obj, float(c_gp.mean())))
        q.grad = free_mask * plan.gq                                     # gradients come from the kernel, not from autograd
        qd.grad = plan.gqd
        opt.step()                                                       # in place: the plan keeps reading q's and qd's buffers
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    coll0 = task.compute_collision(q_start + s * (q_goal - q_start)).any(1).float().mean().item()
    trajs_coll, trajs_free = task.get_trajs_collision_and_free(q, num_interpolation=5)
    n_free = 0 if trajs_free is None else trajs_free.shape[0]
    if verbose:
        for it, c, g in hist:
            print(f"iter {it:4d}: mean collision cost per trajectory {c:9.4f}   mean GP-prior cost {g:9.4f}")
        print(f"{iters} iterations x {batch * horizon} configurations in {elapsed * 1e3:.1f} ms "
              f"({batch * horizon * iters / elapsed:.3g} FK+cost+grad evaluations/s incl. the Python loop)")
        print(f"straight lines in collision: {coll0 * 100:.0f} %   collision-free after optimisation: {n_free}/{batch}")
    return q, n_free, coll0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    main(a.batch, a.horizon, a.iters, a.device)
