import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import gauss_newton_ik as g
from torch_robotics_amd import DifferentiableFrankaPanda, ops
dev="cuda:0"
tree = DifferentiableFrankaPanda(gripper=False, device=dev)
q = torch.rand(4096,7,device=dev)
H = tree.compute_forward_kinematics_all_links(q[:1], link_list=["ee_link"])[0,0]
link = tree._name_to_idx_map["ee_link"]
def t(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:30s} {(time.perf_counter()-t0)/n*1e6:9.1f} us")
t("fk_jacobian", lambda: ops.fk_jacobian(tree._handle, q, None, link))
pos, quat, lin, ang = ops.fk_jacobian(tree._handle, q, None, link)
t("pose_residual", lambda: g.pose_residual(pos, quat, H))
r = g.pose_residual(pos, quat, H)
lam = 1e-4 + 0.1*(r*r).sum(-1)
t("jtj solve", lambda: ops.jtj(lin, ang, r, damping=lam, solve=True))
t("jtj no solve", lambda: ops.jtj(lin, ang, r))
