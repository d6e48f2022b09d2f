import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, torch_robotics_amd as tra
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
qg = q.clone().requires_grad_(True)
def fwd():
    return task.compute_collision_cost(qg)
def fb():
    qg.grad = None
    task.compute_collision_cost(qg).sum().backward()
for _ in range(200): fb()
torch.cuda.synchronize()
# host time of the pieces (GPU never the bound: sync only at the end)
def timeit(name, fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:50s} host {1e6*(t1-t0)/n:7.1f} us   incl. drain {1e6*(time.perf_counter()-t0)/n:7.1f} us")
timeit("forward only (graph kept)", fwd)
c = fwd()
timeit("sum", lambda: c.sum())
s = c.sum()
timeit("backward (retain)", lambda: s.backward(retain_graph=True))
timeit("whole idiom", fb)
timeit("floor fwd", lambda: (qg * 2.0))
m = qg * 2.0; ms = m.sum()
timeit("floor backward (retain)", lambda: ms.backward(retain_graph=True))
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): fwd()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
