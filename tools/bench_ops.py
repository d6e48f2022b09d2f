#!/usr/bin/env python3
"""Secondary benchmark: every C-ABI operator of the unfused drop-in chain at BASELINE config-2 size
(Panda, N = 4096 x 64), timed with HIP events; prints algorithmic bytes / time per op (HBM roofline check).

    python tools/bench_ops.py [--robot panda_arm_no_gripper] [--n 262144]
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

import torch_robotics_amd as tra
from torch_robotics_amd import ops
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
from torch_robotics_amd.kinematics import URDF_DIR
from torch_robotics_amd.kinmodel import KinModel


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robot", default="panda_arm_no_gripper")
    ap.add_argument("--n", type=int, default=4096 * 64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ta = dict(device=dev, dtype=torch.float32)
    kin = KinModel.from_urdf(str(URDF_DIR / f"{args.robot}.urdf"))
    h = ops.ModelHandle(kin)
    L, D, n = kin.n_links, kin.n_dofs, args.n
    q = (torch.rand(n, D, **ta) - 0.5) * 4.0
    rows = []

    def rec(name, nbytes, fn):
        t = timeit(fn)
        rows.append(dict(op=name, us=t * 1e6, bytes_per_sample=nbytes, GBps=nbytes * n / t / 1e9,
                         frac_of_8TBps=nbytes * n / t / 8e12))

    H = ops.fk_forward(h, q)
    gH = torch.randn_like(H)
    pos = ops.fk_positions(h, q)
    gpos = torch.randn_like(pos)
    rec("trk_fk_forward (all links, H)", 4 * D + 64 * L, lambda: ops.fk_forward(h, q))
    rec("trk_fk_forward (EE only)", 4 * D + 64, lambda: ops.fk_forward(h, q, [L - 1]))
    rec("trk_fk_positions (all links)", 4 * D + 12 * L, lambda: ops.fk_positions(h, q))
    rec("trk_fk_backward (gH all links)", 4 * D + 64 * L + 4 * D, lambda: ops.fk_backward(h, q, gH))
    rec("trk_fk_positions_backward", 4 * D + 12 * L + 4 * D, lambda: ops.fk_positions_backward(h, q, gpos))
    qd = torch.randn_like(q)
    rec("trk_fk_jacobian (EE)", 8 * D + 28 + 24 * D, lambda: ops.fk_jacobian(h, q, qd, L - 1))
    rec("trk_rotmat_to_quat (all links)", 64 * L + 16 * L, lambda: ops.rotmat_to_quat(H))
    if args.robot == "panda_arm_no_gripper":
        robot = tra.RobotPanda(tensor_args=ta)
        task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
        Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
        task.set_ee_target(Ht)
        _, cm = task._fused_handles(dev)
        allf = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
        rec("trk_cost_fields (objects, fwd)", 12 * L + 4, lambda: ops.cost_fields(cm, FIELD_OBJECTS, pos))
        rec("trk_cost_fields (all 3, fwd+grad)", 12 * L + 4 + 12 * L, lambda: ops.cost_fields(cm, allf, pos, want_grad=True))
        rec("trk_collision_fields (all 3)", 12 * L + 1, lambda: ops.collision_fields(cm, allf, pos))
        rec("trk_rollout_collision (fused FK + booleans, all 3)", 4 * D + 1, lambda: ops.rollout_collision(h, cm, allf, q))
        rec("trk_rollout_collision (margin 0, objects + ws)", 4 * D + 1, lambda: ops.rollout_collision(h, cm, FIELD_OBJECTS | FIELD_WS, q, margin=0.0))
        # the caller every planner runs after optimisation (tasks.py:234-308): 4096 trajectories x 64 way points, 5 via points
        trajs = q.reshape(4096, -1, D)[:, :64].contiguous() if n >= 4096 * 64 else None
        if trajs is not None:
            n_interp = 4096 * ((64 - 1) * 6 + 1)
            t = timeit(lambda: task.get_trajs_collision_and_free(trajs, return_indices=True), iters=20, warm=3)
            rows.append(dict(op="PlanningTask.get_trajs_collision_and_free (4096 x 64, 5 via points)", us=t * 1e6, bytes_per_sample=4 * D + 1,
                             GBps=(4 * D + 1) * n_interp / t / 1e9, frac_of_8TBps=(4 * D + 1) * n_interp / t / 8e12))
        Hee = H[:, -1].contiguous()
        rec("trk_ee_cost (fwd+grad)", 64 + 4 + 64, lambda: ops.ee_cost(cm, Hee, want_grad=True))
        h.enable_specialized(False)
        rec("trk_rollout_cost_grad (table-driven, c2)", 8 * D + 12 * L + 4, lambda: ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), q))
        h.enable_specialized(True)
        rec("trk_rollout_cost_grad (specialised, c2)", 8 * D + 12 * L + 4, lambda: ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), q))
    for r in rows:
        print(f"{r['op']:44s} {r['us']:9.1f} us  {r['bytes_per_sample']:5d} B/sample  {r['GBps']:8.1f} GB/s  {100 * r['frac_of_8TBps']:5.1f} % of 8 TB/s")
    print(json.dumps({"robot": args.robot, "n": n, "ops": rows}))


if __name__ == "__main__":
    main()
