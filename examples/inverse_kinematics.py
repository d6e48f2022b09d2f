#!/usr/bin/env python3
"""Batched inverse kinematics of the Panda by Adam on the SE(3) distance -- the reference's examples/inverse_kinematics.py
without the matplotlib part: same target construction (`z_rot`, `y_rot`, `Frame`), same `inverse_kinematics` call and return
values.  One iteration is ONE kernel launch (FK + SE3 distance + joint-limit hinge + gradient + Adam update, `trk_ik_step`)
instead of two FK passes, an autograd sweep and ~10 optimiser kernels.  Needs the MI355X: there is no CPU path.

    python examples/inverse_kinematics.py [--batch 10]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch

from torch_robotics_amd import DifferentiableFrankaPanda, Frame, SE3_distance, y_rot, z_rot


def main(batch_size=10, device="cuda:0", max_iters=500, verbose=True):
    torch.manual_seed(0)
    tensor_args = dict(device=device, dtype=torch.float32)
    pos_target = torch.tensor([0.2, 0.4, 0.1], **tensor_args)
    rot_a, rot_b = z_rot(-torch.tensor(torch.pi / 2, **tensor_args)), y_rot(-torch.tensor(torch.pi, **tensor_args))
    rot_target = Frame(rot=rot_a).multiply_transform(Frame(rot=rot_b)).rotation     # z_rot(-pi/2) @ y_rot(-pi)
    frame_target = Frame(rot=rot_target, trans=pos_target, device=device)
    H_target = frame_target.get_transform_matrix()

    diff_panda = DifferentiableFrankaPanda(gripper=False, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q_ik, idx_valid = diff_panda.inverse_kinematics(H_target, link_name="ee_link", batch_size=batch_size, max_iters=max_iters,
                                                    lr=2e-1, se3_eps=5e-2, eps_joint_lim=torch.pi / 64,
                                                    print_freq=50 if verbose else -1, debug=False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    H_ee = diff_panda.compute_forward_kinematics_all_links(q_ik, link_list=["ee_link"])[:, 0]
    err = SE3_distance(H_ee, H_target[0])
    if verbose:
        print(f"\nIK time: {elapsed:.3f} sec")
        print(f"idx_valid: {idx_valid.nelement()}/{batch_size}")
        print("SE(3) distance of the solutions to the target:", [round(float(e), 4) for e in err])
    return q_ik, idx_valid, err, H_target


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    main(a.batch, a.device)
