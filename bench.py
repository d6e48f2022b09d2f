#!/usr/bin/env python3
"""Headline benchmark: FK + cost + gradient rollouts/sec (batch x horizon), Franka Panda 7-DOF.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the fused hot path (`trk_rollout_cost_grad`) over one batch of synthetic joint
trajectories resident in HBM: read q, write link positions, cost and d cost / d q.
Workload (BASELINE.json configs[1], "c2"): batch 4096 x horizon 64, Panda (11 links / 7 DOF), scene
EnvSpheres3D (10 spheres, analytic SDF, cutoff 0.03), cost = object collision + EE SE(3) tracking
(target p=(0.4,0.2,0.5), R=I).  `--config c3` adds self-collision pairs and the workspace box.
Multi-GPU: the batch is sharded, each rank owns 4096 x 64 samples (weak scaling); the only exchange is an
RCCL all-reduce of the packed sums [cost | cost per time step | gradient per time step and joint] (2 kB), issued once per
`--reduce-every` steps on a side stream.

Timed region: W untimed warm-up steps, then exactly K steps bracketed on both sides by a barrier (N > 1: a one-element all-reduce
enqueued on the launch stream -- it completes only when every rank has reached it) + `torch.cuda.synchronize()`; the maximum over the
ranks is taken.  The whole measurement is rehearsed once and discarded first.

Prints ONE JSON line (rank 0).  `roofline.achieved` = 192 algorithmic bytes/sample x samples per launch /
average launch duration (HIP events around the timed region on the launch stream).
`cpu_baseline` = the C oracle (OpenMP over samples, all host cores) on a bounded sample of the same input.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_sample(D, L):
    # read q (4D) + write link positions (12L) + cost (4) + gradient (4D)
    return 4 * D + 12 * L + 4 + 4 * D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="c2", choices=["c2", "c3"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--reduce-every", type=int, default=64)
    ap.add_argument("--graph", type=int, default=0, help="capture this many steps per hipGraph replay (0 = eager launches)")
    ap.add_argument("--weights", default=None, help="experiment: w_self,w_obj,w_ws,w_ee override (reported in config)")
    ap.add_argument("--no-pos", action="store_true", help="experiment: do not write link positions")
    ap.add_argument("--independent-streams", type=int, default=0,
                    help="also report the throughput of unrelated batches alternated over this many HIP streams (secondary "
                         "figure `independent_batches`; off by default so that a rocprofv3 run of the default command sees only "
                         "back-to-back launches of one stream)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="debug: 'gloo' + --single-device lets the N>1 control flow run on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="debug: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import torch_robotics_amd as tra
    from torch_robotics_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    dev = torch.device("cuda", 0 if args.single_device else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    ta = dict(device=dev, dtype=torch.float32)
    robot = tra.RobotPanda(tensor_args=ta)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
    Ht = np.eye(4, dtype=np.float32)
    Ht[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(Ht, w_pos=1.0, w_rot=1.0, square=True)
    weights = (0.0, 1.0, 0.0, 1.0) if args.config == "c2" else (1.0, 1.0, 1.0, 1.0)
    if args.weights:
        weights = tuple(float(v) for v in args.weights.split(","))
    B, H = args.batch, args.horizon
    D, L = robot.q_dim, robot.diff_panda._kin.n_links
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    q = robot.random_q(B * H, generator=gen).reshape(B, H, D).contiguous()
    model, cm = task._fused_handles(dev)
    plan = ops.RolloutPlan(model, cm, weights, q, want_pos=not args.no_pos)
    # the fused kernel leaves one partial cost sum per wavefront (= per trajectory at horizon 64); a rank folds them
    # into a scalar with the deterministic reduce kernel only when a collective needs it
    nb = ops.n_blocks(B * H)
    block_sums = torch.zeros(nb, **ta)
    bs_ptr = block_sums.data_ptr()
    # the collective fires in the MIDDLE of every `reduce_every`-step interval (a planner consumes the sums a few evaluations
    # later), and the interval shrinks for short runs so that the timed region always contains at least one all-reduce
    R = max(1, min(args.reduce_every, args.steps))
    n_slots = 2 * ((args.warmup + args.steps) // R + 2) + 8
    # what a sharded planner exchanges: one packed fp32 buffer [sum cost | sum_b cost(h) | sum_b grad(h, d)]
    packed = torch.zeros((n_slots, 1 + H + H * D), **ta)
    stream = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev) if world > 1 else None

    def reduce_slot(k):
        # sums of the latest evaluation -> one small all-reduce (2 kB), off the launch stream
        buf = packed[k]
        ops.reduce_sum(block_sums, out=buf[0:1])
        torch.sum(plan.cost, dim=0, out=buf[1:1 + H])
        torch.sum(plan.gq, dim=0, out=buf[1 + H:].view(H, D))
        ev = torch.cuda.Event()
        ev.record(stream)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            dist.all_reduce(buf)

    graph = None
    if args.graph > 0:
        # launch-bound inner loop -> hipGraph: G consecutive evaluations per replay (each into its own sum slot
        # of a G-wide window; the window is copied out by the caller when it needs the values)
        G = args.graph
        gstream = torch.cuda.Stream(dev)
        with torch.cuda.stream(gstream):
            for i in range(3):
                plan.launch(bs_ptr, gstream.cuda_stream)
        gstream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=gstream):
            for i in range(G):
                plan.launch(bs_ptr, torch.cuda.current_stream(dev).cuda_stream)

    slot = [0]

    def run(first, count):
        if graph is not None:
            assert count % args.graph == 0
            for _ in range(count // args.graph):
                graph.replay()
            return
        s = stream.cuda_stream
        for j in range(count):
            plan.launch(bs_ptr, s)
            if side is not None and j % R == R // 2:        # counted from the start of this (warm-up or timed) region
                reduce_slot(slot[0])
                slot[0] += 1

    if graph is not None:
        args.steps = max(args.graph, args.steps // args.graph * args.graph)
        args.warmup = max(args.graph, args.warmup // args.graph * args.graph)
    flag = torch.zeros(1, **ta)

    def barrier_in_stream():
        """A barrier that costs no host round trip: a one-element all-reduce enqueued on the launch stream completes on a rank
        only after every rank has reached it, so the `torch.cuda.synchronize()` that follows returns when all ranks are done."""
        dist.all_reduce(flag)

    def measure(first):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize brackets; returns (wall s, event ms)."""
        run(first, args.warmup)
        if world > 1:
            barrier_in_stream()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(first + args.warmup, args.steps)
        ev1.record(stream)
        ta_ = time.perf_counter()
        if world > 1:
            barrier_in_stream()         # closing bracket: in-stream barrier, then ONE synchronize (all streams, all ranks done)
        else:
            while not ev1.query():      # spin on the completion signal: a blocking wait adds ~6 us of wake-up latency
                pass
        tb_ = time.perf_counter()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        if os.environ.get("TRK_BENCH_TRACE"):
            print(f"[trace] submit {1e6 * (ta_ - t0):.1f} us, wait {1e6 * (tb_ - ta_):.1f} us, sync {1e6 * (wall - (tb_ - t0)):.1f} us, "
                  f"events {1e3 * ev0.elapsed_time(ev1):.1f} us", file=sys.stderr)
        return wall, (ev0.elapsed_time(ev1) if graph is None else wall * 1e3)

    # one rehearsal of the whole measurement (discarded): the first pass through the event / sync / launch code paths of a
    # fresh process costs ~10 us more, which matters when the driver asks for only 20 timed steps (~200 us of GPU work)
    measure(0)
    elapsed, ev_ms = measure(args.warmup + args.steps)
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    samples_per_step = B * H * world
    value = samples_per_step * args.steps / elapsed
    bytes_per_launch = algorithmic_bytes_per_sample(D, L) * B * H
    launch_s = ev_ms * 1e-3 / args.steps
    achieved = bytes_per_launch / launch_s / 1e9

    # sanity: the outputs of the last step are finite and the cost sums agree with the per-sample costs
    assert torch.isfinite(plan.cost).all() and torch.isfinite(plan.gq).all()
    last = ops.reduce_sum(block_sums).item()
    ref = plan.cost.double().sum().item()
    assert abs(last - ref) <= 1e-4 * abs(ref) + 1e-3, (last, ref)

    # HBM traffic per launch from the PMC passes recorded under profiles/ (bench.py cannot run rocprofv3 on itself)
    traffic = None
    try:
        key = f"{args.config}:{B}x{H}:{'specialized' if model.specialized else 'table-driven'}"
        if not args.weights and not args.no_pos:
            traffic = json.loads((ROOT / "profiles" / "r02_hbm_traffic.json").read_text())["workloads"][key]["traffic_bytes_per_launch"]
    except Exception:
        traffic = None

    out = {
        "metric": "FK+cost+grad rollouts/sec (batch x horizon), Panda 7-DOF",
        "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: Franka Panda (11 links, 7 DOF), batch={B} x horizon={H} per GPU, "
                               f"fused FK + " + ("SDF-obstacle (EnvSpheres3D, 10 spheres) + EE-tracking"
                                                 if args.config == "c2" else
                                                 "self-collision + SDF-obstacle + workspace box + EE-tracking") +
                               " cost + gradient, q resident in HBM",
                   "global_batch": B * world, "horizon": H, "parallelism": f"batch-sharded x{world}",
                   "launch": "hipGraph x%d" % args.graph if graph is not None else "eager, pre-bound C-ABI call",
                   "kernel": "specialized" if model.specialized else "table-driven",
                   "reduce_every": R if world > 1 else None,
                   **({"experiment_weights": list(weights)} if args.weights else {}),
                   **({"experiment_no_pos": True} if args.no_pos else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_sample": algorithmic_bytes_per_sample(D, L), "launch_us": launch_s * 1e6,
                     # the contract prices against the HBM peak; a launch's working set below the 256 MB Infinity Cache is
                     # absorbed by it (61 - 62 % of the same peak at 400 - 600 MB per launch)
                     "working_set_MB": round(bytes_per_launch / 1e6, 1), "infinity_cache_MB": 256},
    }

    if rank == 0 and world == 1 and graph is None and args.independent_streams > 1:
        # Secondary figure (never `value`): INDEPENDENT batches alternated over three HIP streams.  With one stream a launch waits
        # for the previous one's ~2 us write tail; with three, the next launch's dispatch, kernarg / q fetch and FK overlap it.
        # A planner's iterations depend on each other, so the headline keeps one stream; a server evaluating unrelated batches
        # gets this rate.  Kernels overlap here, so this is a throughput, not a kernel duration.
        ns = args.independent_streams
        streams3 = [torch.cuda.Stream(dev) for _ in range(ns)]
        plans3 = [plan] + [ops.RolloutPlan(model, cm, weights, robot.random_q(B * H, generator=gen).reshape(B, H, D).contiguous(),
                                           want_pos=not args.no_pos) for _ in range(ns - 1)]
        sums3 = [block_sums] + [torch.zeros(nb, **ta) for _ in range(ns - 1)]
        n3 = max(2000, args.steps)

        def run3(n):
            for i in range(n):
                plans3[i % ns].launch(sums3[i % ns].data_ptr(), streams3[i % ns].cuda_stream)
        run3(60)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        run3(n3)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter() - t3
        out["independent_batches"] = {"streams": ns, "steps": n3, "ms_per_step": t3 * 1e3 / n3, "value": B * H * n3 / t3,
                                      "unit": "rollouts/s", "frac_of_hbm_peak": bytes_per_launch * n3 / t3 / 1e9 / HBM_PEAK_GBS,
                                      "note": f"secondary: unrelated batches round-robin over {ns} streams (launches overlap); "
                                              "the headline value above is one stream, launch after launch"}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:       # CPU baseline: rank 0 at N = 1 only
        from oracle import oracle as orc          # checker / baseline only: never on the product path
        o = orc.Oracle(robot.diff_panda._kin, task.build_cost_spec())
        q_host = q.reshape(-1, D).cpu().numpy()
        cores = orc.max_threads()
        probe = q_host[:16384]
        t = time.perf_counter(); o.rollout(probe, weights, "f32"); dt = time.perf_counter() - t
        n_s = int(min(len(q_host), max(16384, len(probe) / dt * args.cpu_seconds / 3)))
        reps, best = 3, 1e30
        for _ in range(reps):
            t = time.perf_counter(); o.rollout(q_host[:n_s], weights, "f32"); best = min(best, time.perf_counter() - t)
        out["cpu_baseline"] = {"value": n_s / best, "unit": "rollouts/s", "cores": cores, "kind": "port",
                               "sample": f"first {n_s} of the {B * H} samples of rank 0's batch, C oracle (fp32, OpenMP "
                                         f"over samples, {cores} threads), best of {reps}"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
