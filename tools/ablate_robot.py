#!/usr/bin/env python3
"""Run-time ablation of a generated rollout kernel (no rebuild): python tools/ablate_robot.py [ur10_allegro|dual_panda|panda]
[first|two] [stamps]   -- `first` / `two`: only the first (two) configuration(s); TRK_ABLATE_LAUNCHES=n timed launches (300); `stamps`: also dump the per-wave phase stamps of that configuration
to gpurun_out/phase_stamps_<robot>_<k>.npy (read them with tools/phase_analyze.py)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D
dev = torch.device("cuda:0")
ident = sys.argv[1] if len(sys.argv) > 1 else "ur10_allegro"
kin, tmpl = codegen.template_for(ident)
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))
spec = CostModelSpec(n_links_in=kin.n_links)
spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
spec.objects = [o.as_object() for o in env.obj_fixed_list]
spec.ee_link = tmpl.ee_link
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
if tmpl.ee2_link >= 0:
    spec.ee2_link = tmpl.ee2_link; spec.ee2_target = Ht
h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
B, H, D, L = 4096, 64, kin.n_dofs, kin.n_links
q = (torch.rand(B, H, D, device=dev) - 0.5) * 3.0
print(f"{ident}: {L} links, {D} DOF, {len(tmpl.obj_links)} collision links, {len(tmpl.self_pairs)} self pairs; {8 * D + 12 * L + 4} B/sample")
for name, w, pos in (("obj+ee, positions", (0, 1, 0, 1), True), ("obj+ee, no positions", (0, 1, 0, 1), False),
                     ("no objectives, positions", (0, 0, 0, 0), True), ("no objectives, no positions", (0, 0, 0, 0), False),
                     ("objects only", (0, 1, 0, 0), True), ("all four", (1, 1, 1, 1), True))[:1 if "first" in sys.argv else (2 if "two" in sys.argv else None)]:
    plan = ops.RolloutPlan(h, cm, w, q, want_pos=pos)
    NL = int(os.environ.get("TRK_ABLATE_LAUNCHES", "300"))
    for _ in range(max(3, NL // 10)): plan.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(NL): plan.launch()
    e1.record(); torch.cuda.synchronize()
    print(f"  {name:30s} {e0.elapsed_time(e1) / NL * 1e3:7.2f} us")
    if "stamps" in sys.argv:
        from torch_robotics_amd._lib import lib
        nb = ops.n_blocks(B * H)
        stamps = torch.zeros((nb, 8), device=dev, dtype=torch.int64)
        lib().trk_debug_set_stamp_buffer(stamps.data_ptr())
        for _ in range(6): plan.launch()
        torch.cuda.synchronize()
        lib().trk_debug_set_stamp_buffer(None)
        out = Path(__file__).resolve().parent.parent / "gpurun_out"; out.mkdir(exist_ok=True)
        tag = name.replace(" ", "_").replace(",", "").replace("+", "_")
        np.save(out / f"phase_stamps_{ident}_{tag}.npy", stamps.cpu().numpy())
