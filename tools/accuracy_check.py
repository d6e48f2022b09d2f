#!/usr/bin/env python3
"""Max deviation of the generated Panda kernel from the fp64 oracle on 65 536 random samples (positions, cost, gradient)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from helpers import gold, model, panda_cost_spec
from oracle.oracle import Oracle
from torch_robotics_amd import ops

g, robot, gs = gold("rollout_panda"), gold("panda_robot"), gold("cost_spheres3d")
m = model("panda_arm_no_gripper")
spec = panda_cost_spec(gs, robot, ee_target=g["target"])
h, cm = ops.ModelHandle(m), ops.CostHandle(spec, "cuda:0")
rng = np.random.default_rng(0)
lo, hi = m.lower[m.controlled], m.upper[m.controlled]
q = (lo + rng.random((65536, 7)) * (hi - lo)).astype(np.float32)
q[::5] *= 1.3                                               # some beyond the limits
pos, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), torch.as_tensor(q, device="cuda:0"))
p64, c64, g64 = Oracle(m, spec).rollout(q.astype(np.float64), (1, 1, 1, 1), "f64")
print(f"specialized={h.specialized}  max |dpos| {np.abs(pos.cpu().numpy() - p64).max():.3e}   "
      f"max |dcost|/max|cost| {np.abs(cost.cpu().numpy() - c64).max() / np.abs(c64).max():.3e}   "
      f"max |dgq|/max|gq| {np.abs(gq.cpu().numpy() - g64).max() / np.abs(g64).max():.3e}")
