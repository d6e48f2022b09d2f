import sys; sys.path.insert(0, ".")
import torch, io, contextlib
import torch_robotics_amd as tra
from torch_robotics_amd.kinematics import DifferentiableFrankaPanda
dev = "cuda:0"
Ht = torch.eye(4, device=dev); Ht[:3, 3] = torch.tensor([0.2, 0.4, 0.1], device=dev)
for check in (1, 20):
  for on in (True, False):
    tree = DifferentiableFrankaPanda(gripper=False, device=dev)
    tree._handle.enable_specialized(on)
    res = []
    for seed in range(6):
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            q, idx = tree.inverse_kinematics(Ht.unsqueeze(0), link_name="ee_link", batch_size=16, max_iters=500, lr=2e-1, se3_eps=5e-2,
                                             eps_joint_lim=torch.pi / 64, print_freq=-1, check_every=check)
        res.append(idx.nelement())
    print("check_every", check, "generated" if on else "table-driven", res)
