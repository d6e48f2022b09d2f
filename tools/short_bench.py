import sys, time, json, subprocess
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); task.set_ee_target(Ht)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
model, cm = task._fused_handles(dev)
plan = ops.RolloutPlan(model, cm, (0, 1, 0, 1), q)
bs = torch.zeros(ops.n_blocks(4096 * 64), **ta); p = bs.data_ptr()
stream = torch.cuda.current_stream(dev); s = stream.cuda_stream
def trial(K, W, prewarm=0, poll=False):
    for _ in range(prewarm): plan.launch(p, s)
    for _ in range(W): plan.launch(p, s)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); ev0.record(stream)
    for _ in range(K): plan.launch(p, s)
    t1 = time.perf_counter()
    ev1.record(stream)
    if poll:
        while not ev1.query():
            pass
    else:
        torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    torch.cuda.synchronize(dev)
    return (t2 - t0) * 1e6 / K, ev0.elapsed_time(ev1) * 1e3 / K, (t1 - t0) * 1e6 / K
print("first trial right after setup (cold clocks):")
for K, W, poll in ((20, 5, False), (20, 5, True), (20, 5, False), (20, 5, True), (20, 5, False), (20, 5, True), (200, 5, False), (200, 5, True)):
    w, e, h = trial(K, W, 0, poll)
    print(f"K={K:5d} W={W:4d} poll={poll!s:5s}: wall {w:6.2f} us/step | events {e:6.2f} us/step | host submit {h:5.2f} us/step")
    time.sleep(0.3)
