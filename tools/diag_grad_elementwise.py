import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import gold, model, panda_cost_spec
from oracle.oracle import Oracle
from torch_robotics_amd import ops
DEV = "cuda:0"
def stats(a, ref, tag):
    a = np.asarray(a, np.float64).reshape(ref.shape[0], -1); ref = np.asarray(ref, np.float64).reshape(ref.shape[0], -1)
    d = np.abs(a - ref); rm = np.abs(ref).max(1, keepdims=True); gm = np.abs(ref).max()
    out = []
    for rtol, ar in ((1e-4, 1e-6), (1e-4, 2e-6), (1e-4, 5e-6), (1e-4, 1e-5), (1e-3, 1e-6), (1e-5, 5e-6)):
        out.append(f"{(d / (rtol * np.abs(ref) + ar * rm + 1e-30)).max():.2f}")
    print(tag, "global", f"{d.max() / gm:.2e}", "elem ratios", out)
g, robot, gs = gold("rollout_panda"), gold("panda_robot"), gold("cost_spheres3d")
m = model("panda_arm_no_gripper")
spec = panda_cost_spec(gs, robot, ee_target=g["target"])
h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
o = Oracle(m, spec)
q = torch.as_tensor(g["q"], device=DEV)
for w, key in (((0, 1, 0, 1), "gq_c2"), ((1, 1, 1, 1), "gq_c3")):
    for on in (True, False):
        h.enable_specialized(on)
        _, c, gq = ops.rollout_cost_grad(h, cm, w, q)
        stats(gq.cpu().numpy().reshape(-1, 7), g[key].reshape(-1, 7), f"golden {key} spec={on}")
rng = np.random.default_rng(3)
for n in (1000, 20000):
    qq = rng.uniform(-3.0, 3.9, (n, 7)).astype(np.float32)
    for w in ((1, 1, 1, 1), (0, 1, 0, 1), (1, 0, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)):
        _, c64, g64 = o.rollout(qq.astype(np.float64), w, "f64")
        _, c32, g32 = o.rollout(qq, w, "f32")
        for on in (True, False):
            h.enable_specialized(on)
            _, c, gq = ops.rollout_cost_grad(h, cm, w, torch.as_tensor(qq, device=DEV))
            stats(gq.cpu().numpy(), g64, f"rand n={n} w={w} spec={on}")
        stats(g32, g64, f"  (oracle f32 vs f64) n={n} w={w}")
