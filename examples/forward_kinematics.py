#!/usr/bin/env python3
"""Batched forward kinematics of six robots -- the plumbing of the reference's examples/forward_kinematics.py with
`torch_robotics` replaced by `torch_robotics_amd` (same class names, methods and tensor layouts; the compute is one HIP kernel
launch per call instead of ~117 batched matmuls).  Needs the MI355X: there is no CPU COMPUTE path -- `--device cpu` (the reference
example's own setting, examples/forward_kinematics.py:15) keeps the tensors on the host and copies them to the GPU and back around
every call (differentiably), so the reference's script runs with only its imports changed.

    python examples/forward_kinematics.py [--batch 10] [--device cpu|cuda:0]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch

from torch_robotics_amd import (DifferentiableAllegroHand, DifferentiableFrankaPanda, DifferentiableHabitatStretch,
                                DifferentiableShadowHand, DifferentiableTiagoDualHoloMove, DifferentiableUR10,
                                link_pos_from_link_tensor, link_quat_from_link_tensor)


def main(batch_size=10, device="cuda:0", verbose=True):
    torch.manual_seed(1)
    results = {}
    for title, cls in (("Panda", DifferentiableFrankaPanda), ("UR10", DifferentiableUR10),
                       ("Habitat Stretch", DifferentiableHabitatStretch), ("Tiago", DifferentiableTiagoDualHoloMove),
                       ("Shadow Hand", DifferentiableShadowHand), ("Allegro Hand", DifferentiableAllegroHand)):
        model = cls(device=device)
        if verbose:
            print(f"\n=========================== {title} Model ===============================")
            model.print_link_names()
            print(model.get_joint_limits())
            print(model._n_dofs)
        q = torch.rand(batch_size, model._n_dofs).to(device).requires_grad_(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = model.compute_forward_kinematics_all_links(q)          # (batch, links, 4, 4), differentiable w.r.t. q
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        pos, quat = link_pos_from_link_tensor(data), link_quat_from_link_tensor(data)
        data[..., :3, 3].sum().backward()                             # explicit reverse-mode kernel behind autograd
        if verbose:
            print(tuple(data.shape), tuple(pos.shape), tuple(quat.shape))
            print(f"Computational Time {elapsed:.4f}")
        results[title] = (data.detach(), q.grad.detach())
    return results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    main(a.batch, a.device)
