#!/usr/bin/env python3
"""A batch of trajectories sharded over the GPUs of one node: every rank evaluates its block with the fused FK + cost + gradient
kernel and the ranks exchange only the packed sums (total cost, cost per time step, gradient per time step and joint: 2 kB) through
the peer-to-peer mailbox -- SURVEY.md 8e.  The reference is single-device; this is what its `PlanningTask` batch looks like sharded.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 examples/sharded_rollouts.py
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 examples/sharded_rollouts.py --single-device --backend gloo
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL's and the mailbox's peer mappings) on this driver

import torch
import torch.distributed as dist

import torch_robotics_amd as tra
from torch_robotics_amd.distributed import ShardedRollout, shard_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768, help="trajectories over ALL ranks (BASELINE configs[2]: 32768 x 64)")
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true", help="debug: every rank uses cuda:0")
    a = ap.parse_args()
    dev = torch.device("cuda", 0 if a.single_device else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))
    rank, world = dist.get_rank(), dist.get_world_size()
    ta = dict(device=dev, dtype=torch.float32)
    robot = tra.RobotPanda(tensor_args=ta)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, clamp_sdf=True, tensor_args=ta)
    lo, hi = shard_batch(a.batch, rank, world)                      # contiguous blocks of whole trajectories
    q = robot.random_q((hi - lo) * a.horizon, generator=torch.Generator(device=dev).manual_seed(1234 + rank))
    q = q.reshape(hi - lo, a.horizon, robot.q_dim).contiguous()
    plan = task.rollout_plan(q, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False)
    sh = ShardedRollout(plan)                                       # collective: mailbox when it validates, else the all-reduce
    H, D = a.horizon, robot.q_dim
    lr = 1e-3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.iters):
        sh.launch()                                                 # this rank's cost (B/N, H) and gradient (B/N, H, D)
        sh.send()                                                   # packed sums -> every peer; never waits
        q.sub_(lr * plan.gq)                                        # the rank's own descent step on its block (reads plan.gq, writes q in place)
        total = sh.recv()                                           # sums over ALL ranks; the same bits on every rank
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        cost_sum = float(total[0]); g_norm = float(total[1 + H:].norm())
        print(f"{world} ranks x {hi - lo} trajectories x {H}: {a.iters} iterations in {dt * 1e3:.1f} ms "
              f"({a.batch * H * a.iters / dt:.3g} FK+cost+grad evaluations/s incl. the Python loop); exchange via "
              f"{'the peer-to-peer mailbox' if sh.mailbox is not None else 'all-reduce (' + str(sh.mailbox_note) + ')'}; "
              f"global hinge cost {cost_sum:.4g}, |sum_b gradient| {g_norm:.4g}")
    dist.barrier()
    sh.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
