import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, numpy as np
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5); task.set_ee_target(T)
for B in (4096, 64):
    q = robot.random_q(B * 64).reshape(B, 64, 7).contiguous()
    plan = task.rollout_plan(q, w_self=0, w_ws=0, w_ee=1.0)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(100): plan.launch(None, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): plan.launch(None, s)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host {1e6*(t1-t0)/2000:.2f} us per launch call, total incl. drain {1e6*(t2-t0)/2000:.2f} us/step")
