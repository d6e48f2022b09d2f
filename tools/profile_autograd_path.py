#!/usr/bin/env python3
"""Where the time of `task.compute_collision_cost(q).sum().backward()` goes at 4096 x 64: forward with grad 17 us, + sum 27 us,
+ backward 94 us -- the autograd engine's hand-off, not the kernels (fused kernel 9.5 us, the backward's one multiply 5.7 us)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, torch_robotics_amd as tra
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
qg = q.clone().requires_grad_(True)
def t(name, fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")
t("forward with grad", lambda: task.compute_collision_cost(qg))
t("forward + sum", lambda: task.compute_collision_cost(qg).sum())
def fb():
    qg.grad = None
    task.compute_collision_cost(qg).sum().backward()
t("forward + sum + backward", fb)
ones = torch.ones(4096, 64, **ta)
def fb2():
    qg.grad = None
    task.compute_collision_cost(qg).backward(ones)
t("forward + backward(ones)", fb2)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20): fb()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
