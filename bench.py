#!/usr/bin/env python3
"""Headline benchmark: FK + cost + gradient rollouts/sec (batch x horizon), Franka Panda 7-DOF.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this script starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py ...` itself as a CHILD process, before anything in this process has touched the GPU, and
exits with the child's code.  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one rank.

One step = one pass of the fused hot path (`trk_rollout_cost_grad`) over one batch of synthetic joint
trajectories resident in HBM: read q, write link positions, cost and d cost / d q.
Workload (BASELINE.json configs[1], "c2"): batch 4096 x horizon 64, Panda (11 links / 7 DOF), scene
EnvSpheres3D (10 spheres, analytic SDF, cutoff 0.03), cost = object collision + EE SE(3) tracking
(target p=(0.4,0.2,0.5), R=I).  `--config c3` adds self-collision pairs and the workspace box (configs[2]'s objective stack).
`--scene grid|shelf|maze` are SURVEY 8(d)'s secondary runs of the same workload: the 200^3 voxel SDF of the same spheres
(+16 B of gathers per collision link and sample), EnvTableShelf and EnvMazeBoxes3D (box scenes; same algorithmic bytes).
Multi-GPU: the batch is sharded, each rank owns 4096 x 64 samples (weak scaling); the only exchange is an
RCCL all-reduce of the packed sums [cost | cost per time step | gradient per time step and joint] (2 kB), issued once per
`--reduce-every` steps on a side stream.  `value` INCLUDES those collectives; `multi_gpu.kernel_only` is the same loop without them,
`multi_gpu.full_stack_c3` is configs[2]'s objective stack on the same shards.

Timed region: W untimed warm-up steps, then exactly K steps bracketed on both sides by a barrier (N > 1: a one-element all-reduce
enqueued on the launch stream -- it completes only when every rank has reached it) + `torch.cuda.synchronize()`.  A rank's clock
runs from the opening bracket to the return of its own closing `torch.cuda.synchronize()` (launch stream AND the side stream's
collectives complete); the maximum over the ranks is taken -- the time from the common start to the completion of the last rank.
For N > 1 the closing barrier follows the clock stop (an 8-rank all-reduce is not a step; at N = 1 there is none either).  The whole measurement is rehearsed once and discarded first.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes/sample x samples per launch /
average launch duration (HIP events around the timed region on the launch stream).
`cpu_baseline` = the C oracle (OpenMP over samples) on a bounded sample of the same input, on all host threads (`value`) and on one.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# fp32 vector peak 157.3 TFLOP/s = 1024 SIMDs x 2.4 GHz x 32 lanes x 2 flop: one 64-lane VALU instruction takes a SIMD 2 cycles
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2
PMC_FILES = ("r03_pmc.json", "r02_hbm_traffic.json")      # per-launch counters recorded by tools/run_r03_profiles.sh


def algorithmic_bytes_per_sample(D, L, n_grid_links=0):
    # read q (4D) + write link positions (12L) + cost (4) + gradient (4D); the voxel-grid scene adds one
    # 16-byte gather (sdf + stored gradient of the nearest cell) per collision link
    return 4 * D + 12 * L + 4 + 4 * D + 16 * n_grid_links


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="c2", choices=["c2", "c3"])
    ap.add_argument("--scene", default="spheres", choices=["spheres", "grid", "shelf", "maze"])
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
    ap.add_argument("--same-q", action="store_true", help="debug: every rank draws the same q (all-reduced sums == N x rank 0's)")
    ap.add_argument("--native-rccl", action="store_true",
                    help="issue the planner's exchange straight on librccl (torch_robotics_amd.distributed.RcclAllReduce: one ctypes "
                         "call per collective) instead of torch.distributed.all_reduce; the barriers stay on torch.distributed")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: run the N > 1 code path (process group, side-stream all-reduce, in-stream barriers, multi_gpu section) "
                         "even with ONE rank -- the only way to put the RCCL calls on hardware on a 1-GPU box")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N > 1 and no launcher around us: start the N ranks as a child `torch.distributed.run` and return its exit code.
    Nothing in this process has imported torch or touched the GPU yet (a process that has must never exec / be replaced)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")                     # torchrun would set 1; the ranks do no CPU arithmetic anyway
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def cpu_model_name():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_record(key):
    """Per-launch PMC figures of a workload from profiles/ (bench.py cannot run rocprofv3 on itself): (record, file name)."""
    for name in PMC_FILES:
        try:
            rec = json.loads((ROOT / "profiles" / name).read_text())["workloads"].get(key)
        except (OSError, ValueError, KeyError):
            rec = None
        if rec:
            return rec, "profiles/" + name
    return None, None


def make_task(tra, scene, ta):
    import numpy as np
    if scene == "spheres":
        env, what = tra.EnvSpheres3D(tensor_args=ta), "EnvSpheres3D, 10 spheres, analytic SDF"
    elif scene == "grid":
        env = tra.EnvSpheres3D(precompute_sdf_obj_fixed=True, sdf_cell_size=0.01, tensor_args=ta)
        what = "EnvSpheres3D as a 200^3 voxel SDF with stored gradients (GridMapSDF, cell 0.01)"
    elif scene == "shelf":
        env, what = tra.EnvTableShelf(tensor_args=ta), "EnvTableShelf, 1 + 10 rounded boxes in 2 posed objects"
    else:
        env, what = tra.EnvMazeBoxes3D(tensor_args=ta), "EnvMazeBoxes3D, 14 rounded boxes"
    robot = tra.RobotPanda(tensor_args=ta)
    task = tra.PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
    Ht = np.eye(4, dtype=np.float32)
    Ht[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(Ht, w_pos=1.0, w_rot=1.0, square=True)
    return robot, task, what


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # a rank started by an outside launcher: dmabuf IPC for RCCL's peer mappings
    import numpy as np
    import torch
    import torch.distributed as dist
    import torch_robotics_amd as tra
    from torch_robotics_amd import ops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = torch.device("cuda", 0 if args.single_device else local_rank)
    torch.cuda.set_device(dev)
    distributed = world > 1 or args.force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                     # --force-dist without a launcher: a one-rank group of our own
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    ta = dict(device=dev, dtype=torch.float32)
    robot, task, scene_text = make_task(tra, args.scene, ta)
    W_C2, W_C3 = (0.0, 1.0, 0.0, 1.0), (1.0, 1.0, 1.0, 1.0)
    weights = W_C2 if args.config == "c2" else W_C3
    if args.weights:
        weights = tuple(float(v) for v in args.weights.split(","))
    B, H = args.batch, args.horizon
    D, L = robot.q_dim, robot.diff_panda._kin.n_links
    gen = torch.Generator(device=dev).manual_seed(1234 + (0 if args.same_q else rank))
    q = robot.random_q(B * H, generator=gen).reshape(B, H, D).contiguous()
    model, cm = task._fused_handles(dev)
    plan = ops.RolloutPlan(model, cm, weights, q, want_pos=not args.no_pos)
    # the fused kernel leaves one partial cost sum per wavefront (= per trajectory at horizon 64); a rank folds them
    # into a scalar with the deterministic reduce kernel only when a collective needs it
    nb = ops.n_blocks(B * H)
    block_sums = torch.zeros(nb, **ta)
    bs_ptr = block_sums.data_ptr()
    # The planner's cadence: one exchange per `reduce_every` evaluations, counted over the launches of a measurement (warm-up included),
    # fired in the middle of its interval (the sums are consumed a few evaluations later).  The cadence does not shrink for short
    # runs: a 20-step region of a 64-step cadence contains a collective with probability 20 / 64, here deterministically by the
    # launch count -- `multi_gpu.collectives_in_timed_region` says how many it was, `multi_gpu.exchange_us` what one costs.
    R = max(1, args.reduce_every)
    n_slots = 8 * ((args.warmup + args.steps) // R + 2) + 16
    # what a sharded planner exchanges: one packed fp32 buffer [sum cost | sum_b cost(h) | sum_b grad(h, d)]
    packed = torch.zeros((n_slots, 1 + H + H * D), **ta)
    stream = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev) if distributed else None

    packers = {}

    def pack_sums(pl, buf):
        """The sums of the latest evaluation of plan `pl` -> buf: one launch of trk_pack_sums on the launch stream."""
        pk = packers.get(id(pl))
        if pk is None:
            pk = packers[id(pl)] = ops.PackedSums(pl, block_sums)
        pk.pack(buf, stream.cuda_stream)

    native = None
    if distributed and args.native_rccl:
        from torch_robotics_amd.distributed import RcclAllReduce
        native = RcclAllReduce(dev)

    def reduce_slot(pl, k):
        # sums of the latest evaluation -> one small all-reduce (2 kB), off the launch stream
        buf = packed[k]
        pack_sums(pl, buf)
        ev = torch.cuda.Event()
        ev.record(stream)
        side.wait_event(ev)
        if native is not None:
            native.all_reduce_sum_(buf, side.cuda_stream)
        else:
            with torch.cuda.stream(side):
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
    launches = [0]          # launches of this process so far: the cadence counter

    def run(pl, count, collectives):
        if graph is not None:
            assert count % args.graph == 0
            for _ in range(count // args.graph):
                graph.replay()
            return
        s = stream.cuda_stream
        for j in range(count):
            pl.launch(bs_ptr, s)
            launches[0] += 1
            if collectives and launches[0] % R == R // 2:
                reduce_slot(pl, slot[0] % n_slots)
                slot[0] += 1

    if graph is not None:
        args.steps = max(args.graph, args.steps // args.graph * args.graph)
        args.warmup = max(args.graph, args.warmup // args.graph * args.graph)
    flag = torch.zeros(1, **ta)

    def barrier_in_stream():
        """A barrier that costs no host round trip: a one-element all-reduce enqueued on the launch stream completes on a rank
        only after every rank has reached it, so the `torch.cuda.synchronize()` that follows returns when all ranks are done."""
        dist.all_reduce(flag)

    n_coll = [0]            # collectives inside the latest timed region

    def measure(pl, collectives):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize brackets; returns (wall s, event ms),
        the wall time already as the maximum over the ranks."""
        launches[0] = 0             # the cadence is counted from the first warm-up step of THIS measurement
        run(pl, args.warmup, collectives)
        if distributed:
            barrier_in_stream()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        slots_before = slot[0]
        run(pl, args.steps, collectives)
        n_coll[0] = slot[0] - slots_before
        ev1.record(stream)
        ev_side = None
        if side is not None and collectives:
            ev_side = torch.cuda.Event()
            ev_side.record(side)        # the collectives issued inside the region belong to it
        ta_ = time.perf_counter()
        # The clock of a rank stops when ITS K steps (and its side-stream collectives) are complete and its closing synchronize has
        # returned; every rank started from a common barrier, so the MAX over ranks below is the time from that barrier to the
        # completion of the last rank -- what a closing barrier of zero latency would measure.  The closing barrier itself comes
        # after the clock stop: an 8-rank all-reduce over xGMI is tens of microseconds, a fifth of the region at 20 timed steps.
        while not ev1.query() or (ev_side is not None and not ev_side.query()):
            pass                        # spin on the completion signal: a blocking wait adds ~6 us of wake-up latency
        tb_ = time.perf_counter()
        torch.cuda.synchronize(dev)     # the closing synchronize (this rank's streams are idle by now)
        wall = time.perf_counter() - t0
        if distributed:
            barrier_in_stream()         # the closing barrier, behind the clock stop: every rank has finished its region
            torch.cuda.synchronize(dev)
        if os.environ.get("TRK_BENCH_TRACE"):
            print(f"[trace] submit {1e6 * (ta_ - t0):.1f} us, wait {1e6 * (tb_ - ta_):.1f} us, closing bracket "
                  f"{1e6 * (time.perf_counter() - tb_):.1f} us, events {1e3 * ev0.elapsed_time(ev1):.1f} us", file=sys.stderr)
        ev_ms = ev0.elapsed_time(ev1) if graph is None else wall * 1e3
        if distributed:
            tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall, ev_ms

    # one rehearsal of the whole measurement (discarded): the first pass through the event / sync / launch code paths of a
    # fresh process costs ~10 us more, which matters when the driver asks for only 20 timed steps (~200 us of GPU work)
    measure(plan, distributed)
    elapsed, ev_ms = measure(plan, distributed)
    n_coll_value = n_coll[0]

    samples_per_step = B * H * world
    value = samples_per_step * args.steps / elapsed
    n_grid_links = len(cm.spec.obj_link_idx) if (cm.spec.grid is not None and weights[1] != 0.0) else 0
    bps = algorithmic_bytes_per_sample(D, L, n_grid_links)
    bytes_per_launch = bps * B * H
    launch_s = ev_ms * 1e-3 / args.steps
    achieved = bytes_per_launch / launch_s / 1e9

    # sanity: the outputs of the last step are finite and the cost sums agree with the per-sample costs
    assert torch.isfinite(plan.cost).all() and torch.isfinite(plan.gq).all()
    last = ops.reduce_sum(block_sums).item()
    ref = plan.cost.double().sum().item()
    assert abs(last - ref) <= 1e-4 * abs(ref) + 1e-3, (last, ref)

    # PMC figures of this workload recorded under profiles/ (separate rocprofv3 --pmc passes of this very command)
    kind = "specialized" if model.specialized else "table-driven"
    pmc_key = f"{args.config}:{args.scene}:{B}x{H}:{kind}"
    pmc, pmc_src = (None, None) if (args.weights or args.no_pos) else pmc_record(pmc_key)
    if pmc is None and args.scene == "spheres" and not (args.weights or args.no_pos):
        pmc, pmc_src = pmc_record(f"{args.config}:{B}x{H}:{kind}")                  # round-2 key format
    traffic = pmc.get("traffic_bytes_per_launch") if pmc else None
    valu_per_wave = pmc.get("valu_insts_per_wave") if pmc else None
    if valu_per_wave is None and pmc_src and args.scene == "spheres" and args.config == "c2":
        valu_per_wave = 911.0                                                       # profiles/r02_pmc_sq_raw.txt
    valu_frac = None if valu_per_wave is None else valu_per_wave * nb / launch_s / VALU_PEAK_WAVE_INSTS_PER_S

    objectives = {"c2": "SDF-obstacle + EE-tracking", "c3": "self-collision + SDF-obstacle + workspace box + EE-tracking"}
    out = {
        "metric": "FK+cost+grad rollouts/sec (batch x horizon), Panda 7-DOF",
        "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: Franka Panda (11 links, 7 DOF), batch={B} x horizon={H} per GPU, "
                               f"fused FK + {objectives[args.config]} cost + gradient ({scene_text}), q resident in HBM",
                   "global_batch": B * world, "horizon": H, "parallelism": f"batch-sharded x{world}",
                   "scene": args.scene, "objectives": args.config,
                   "launch": "hipGraph x%d" % args.graph if graph is not None else "eager, pre-bound C-ABI call",
                   "kernel": kind,
                   "reduce_every": R if distributed else None,
                   **({"experiment_weights": list(weights)} if args.weights else {}),
                   **({"experiment_no_pos": True} if args.no_pos else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_src if traffic else None,
                     "bytes_per_sample": bps, "launch_us": launch_s * 1e6,
                     # second bound: VALU instructions per wavefront (SQ_INSTS_VALU / SQ_WAVES of the same command)
                     # x wavefronts per launch / launch time, against one 64-lane instruction per 2 cycles per SIMD
                     # (= the 157.3 TFLOP/s fp32 vector peak when every instruction is an FMA)
                     "valu_frac": valu_frac, "valu_insts_per_wave": valu_per_wave,
                     "valu_source": pmc_src if valu_per_wave else None,
                     # the contract prices against the HBM peak; a launch's working set below the 256 MB Infinity Cache is
                     # absorbed by it (61 - 62 % of the same peak at 400 - 600 MB per launch)
                     "working_set_MB": round(bytes_per_launch / 1e6, 1), "infinity_cache_MB": 256},
    }

    if distributed:
        # (1) the same loop without the collectives; (2) configs[2]'s objective stack on the same shards, with collectives;
        # (3) a check of the exchange itself: all-reduced packed sums == the sum of the ranks' local packed sums
        ko_elapsed, _ = measure(plan, False)
        plan3 = plan if weights == W_C3 else ops.RolloutPlan(model, cm, W_C3, q, want_pos=not args.no_pos)
        if plan3 is not plan:
            measure(plan3, True)
        c3_elapsed, c3_ev = (elapsed, ev_ms) if plan3 is plan else measure(plan3, True)
        # what ONE exchange costs end to end when nothing hides it: pack kernel + all-reduce, from enqueue to completion
        torch.cuda.synchronize(dev)
        t_x = time.perf_counter()
        for k in range(10):
            reduce_slot(plan, (slot[0] + k) % n_slots)
            side.synchronize()
        exchange_us = (time.perf_counter() - t_x) / 10 * 1e6
        plan.launch(bs_ptr, stream.cuda_stream)
        local = torch.zeros(1 + H + H * D, **ta)
        pack_sums(plan, local)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        reduced = local.clone()
        if native is not None:
            native.all_reduce_sum_(reduced, stream.cuda_stream)
        else:
            dist.all_reduce(reduced)
        torch.cuda.synchronize(dev)
        expect = torch.stack(gathered).double().sum(0)
        err = float(((reduced.double() - expect).abs() / (expect.abs() + 1.0)).max().item())
        out["multi_gpu"] = {
            "backend": dist.get_backend(), "ranks": dist.get_world_size(), "reduce_every": R,
            "exchange_via": "librccl ncclAllReduce (ctypes)" if native is not None else "torch.distributed.all_reduce",
            "collectives_in_timed_region": n_coll_value, "exchange_us": exchange_us,
            "allreduce_floats": int(local.numel()),
            "with_allreduce": {"value": value, "ms_per_step": elapsed * 1e3 / args.steps},
            "kernel_only": {"value": samples_per_step * args.steps / ko_elapsed, "ms_per_step": ko_elapsed * 1e3 / args.steps},
            "full_stack_c3": {"value": samples_per_step * args.steps / c3_elapsed, "ms_per_step": c3_elapsed * 1e3 / args.steps,
                              "launch_us": c3_ev * 1e3 / args.steps,
                              "workload": "BASELINE configs[2] objective stack (self-collision + SDF-obstacle + workspace box + "
                                          f"EE) on the same shards: {B * world} x {H} over {world} GPUs, with the all-reduce"},
            "allreduce_check": {"ok": bool(err < 1e-5), "max_rel_err": err,
                                "sum_cost_all_ranks": float(reduced[0].item()), "sum_cost_rank0": float(gathered[0][0].item())},
        }
        assert err < 1e-5, f"all-reduced sums differ from the sum of the ranks' sums: {err}"

    if rank == 0 and not distributed and graph is None and args.independent_streams > 1:
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

    if rank == 0 and not distributed and args.cpu_seconds > 0:       # CPU baseline: rank 0 at N = 1 only
        from oracle import oracle as orc          # checker / baseline only: never on the product path
        spec = task.build_cost_spec()
        if spec.grid is not None:                 # the oracle reads host arrays
            spec.grid = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in spec.grid.items()}
        o = orc.Oracle(robot.diff_panda._kin, spec)
        q_host = q.reshape(-1, D).cpu().numpy()
        cores = orc.max_threads()

        def timed(n, reps):
            best = 1e30
            for _ in range(reps):
                t = time.perf_counter(); o.rollout(q_host[:n], weights, "f32"); best = min(best, time.perf_counter() - t)
            return best
        # all threads: ~2/3 of the budget; one thread: ~1/3 (both bounded samples of rank 0's batch)
        probe = min(16384, len(q_host))
        dt = timed(probe, 1)
        n_all = int(min(len(q_host), max(probe, probe / dt * args.cpu_seconds * 2 / 9)))
        t_all = timed(n_all, 3)
        orc.set_threads(1)
        dt1 = timed(2048, 1)
        n_one = int(min(len(q_host), max(2048, 2048 / dt1 * args.cpu_seconds / 6)))
        t_one = timed(n_one, 2)
        orc.set_threads(cores)
        out["cpu_baseline"] = {"value": n_all / t_all, "unit": "rollouts/s", "cores": cores, "kind": "port",
                               "cpu_model": cpu_model_name(),
                               "sample": f"first {n_all} of the {B * H} samples of rank 0's batch, C oracle (fp32, OpenMP "
                                         f"over samples, {cores} threads), best of 3",
                               "one_core": {"value": n_one / t_one, "unit": "rollouts/s", "cores": 1,
                                            "sample": f"first {n_one} samples, same code on 1 thread, best of 2"}}
    elif rank == 0:
        out["cpu_baseline"] = None

    if distributed:
        if native is not None:
            torch.cuda.synchronize(dev)
            native.close()
        dist.barrier()
        dist.destroy_process_group()
    # The JSON line is the LAST thing on stdout: librccl announces itself with a printf ("Librccl path : ...") that sits in the
    # C stdio buffer until it is flushed -- without this it would land behind the line at process exit.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
