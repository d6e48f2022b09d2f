#!/usr/bin/env python3
"""trk_ik_gn_steps: time per Gauss-Newton iteration (K iterations per launch) against the two-launch form, Panda."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "examples"))
import torch
from torch_robotics_amd import DifferentiableFrankaPanda, ops
import gauss_newton_ik as ex

dev = "cuda:0"
tree = DifferentiableFrankaPanda(gripper=False, device=dev)
lo, hi, _, _ = tree.get_joint_limit_array()
lo, hi = (torch.as_tensor(a, device=dev, dtype=torch.float32) for a in (lo, hi))
link = tree._name_to_idx_map["ee_link"]
torch.manual_seed(0)
H = tree.compute_forward_kinematics_all_links(lo + torch.rand(1, 7, device=dev) * (hi - lo), link_list=["ee_link"])[0, 0].contiguous()
for n in (4096, 65536, 262144):
    q0 = (lo + torch.rand(n, 7, device=dev) * (hi - lo)).contiguous()
    for K in (1, 10, 32):
        q = q0.clone()
        ops.ik_gn_steps(tree._handle, link, H, lo, hi, q, K)
        torch.cuda.synchronize()
        reps = max(3, 200 // K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.ik_gn_steps(tree._handle, link, H, lo, hi, q, K)
        e1.record(); torch.cuda.synchronize()
        print(f"n={n:7d}  K={K:2d} iterations per launch: {e0.elapsed_time(e1) / reps / K * 1e3:8.2f} us per iteration")
    q = q0.clone()
    t0 = time.perf_counter()
    for _ in range(10):
        pos, quat, lin, ang = ops.fk_jacobian(tree._handle, q, None, link)
        r = ex.pose_residual(pos, quat, H)
        _, _, dq = ops.jtj(lin, ang, r, damping=1e-4 + 0.1 * (r * r).sum(-1), solve=True)
        q = torch.minimum(torch.maximum(q + dq, lo), hi)
    torch.cuda.synchronize()
    print(f"n={n:7d}  two-launch form (trk_fk_jacobian + trk_jtj + torch ops): {(time.perf_counter() - t0) / 10 * 1e6:8.1f} us per iteration")
