#!/usr/bin/env python3
"""Host-side profile of PlanningTask.get_trajs_collision_and_free (4096 x 64, 5 via points): where the Python microseconds of a call go
(cProfile over 2000 calls; the GPU work of a call is ~35 us, the call ~57).  usage: tools/profile_traj_validation_host.py"""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch_robotics_amd as tra

dev = torch.device("cuda:0")
TA = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=TA)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
for _ in range(50):
    task.get_trajs_collision_and_free(q, num_interpolation=5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    task.get_trajs_collision_and_free(q, num_interpolation=5)
print(f"{(time.perf_counter() - t0) / 500 * 1e6:.1f} us per call (no profiler)")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    task.get_trajs_collision_and_free(q, num_interpolation=5)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
