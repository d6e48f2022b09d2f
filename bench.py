#!/usr/bin/env python3
"""Headline benchmark: FK + cost + gradient rollouts/sec (batch x horizon) -- every BASELINE config as a driver-runnable workload.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

N > 1 without a launcher: this script starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 --master-port P bench.py ...` itself as a CHILD process, before anything in this process has touched the GPU, and
exits with the child's code.  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one rank.

One step = one pass of the hot path over one batch of synthetic joint trajectories resident in HBM.
  c2 (default; BASELINE configs[1], the headline): Panda (11 links / 7 DOF), batch 4096 x horizon 64, EnvSpheres3D (10 spheres, analytic
      SDF, cutoff 0.03), object collision + EE SE(3) tracking (target p=(0.4,0.2,0.5), R=I): ONE launch of `trk_rollout_cost_grad`
      (read q, write link positions, cost and d cost / d q).  `--scene grid|shelf|maze` are SURVEY 8(d)'s secondary runs.
  c3 (configs[2]'s objective stack): the same + self-collision pairs + workspace box; 32768 x 64 is 8 ranks x this.
  c4 (configs[3]): UR10 + Allegro hand (30 links / 22 DOF), 4096 x 64, "FK + Jacobian + cost": the fused rollout (FK, obstacle + EE
      cost, gradient, link positions) and the geometric Jacobian of `ee_link` in ONE launch (`trk_rollout_jacobian_cost_grad`: the
      columns are read out of the poses the rollout holds; `--two-launch` = the rollout, then `trk_fk_jacobian`: rounds 2 - 5).
  c5 (configs[4]): dual Panda (23 links / 14 DOF), horizon 128, fp16 q / qd / link positions / gradients in HBM with fp32 arithmetic
      and cost, GP-smoothness (sigma_gp = 0.1, dt = 5 / 128) + obstacle + EE on both arms; 2048 trajectories per GPU -- at
      `--gpus 4` the global batch is BASELINE's 8192.  ONE launch per step (`trk_rollout_gp_cost_grad`: q / qd read once, gq / gqd
      written once; `--two-launch` = the rollout, then the prior accumulated into its gradient).  The gradients carry a
      power-of-two loss scale (`ops.gp_grad_scale`).
Multi-GPU: the batch is sharded, each rank owns its block of whole trajectories (weak scaling); the only exchange is an
all-reduce (RCCL) of the packed sums [cost | cost per time step | gradient per time step and joint] (1 + H + H D floats), issued
on a side stream once per `--reduce-every` steps -- at most `steps`, so EVERY timed region contains at least one exchange
(`multi_gpu.collectives_in_timed_region`).  `value` INCLUDES those collectives; `multi_gpu.kernel_only` is the same loop without
them, `multi_gpu.every_step` the same loop with one exchange per step, `multi_gpu.full_stack_c3` (c2 / c3 runs) configs[2]'s
objective stack on the same shards.

Timed region: W untimed warm-up steps, then exactly K steps bracketed on both sides by a barrier (N > 1: a one-element all-reduce
enqueued on the launch stream -- it completes only when every rank has reached it) + `torch.cuda.synchronize()`.  A rank's clock
runs from the opening bracket to the return of its own closing `torch.cuda.synchronize()` (launch stream AND the side stream's
collectives complete); the maximum over the ranks is taken -- the time from the common start to the completion of the last rank.
For N > 1 the closing barrier FOLLOWS the clock stop (it is outside the clock: an 8-rank all-reduce is not a step; at N = 1 there
is none either).  The whole measurement is rehearsed once and discarded first.

Launch mode: the step loop is captured -- every run of consecutive steps between two exchanges is replayed as hipGraphs of at most
`--graph` (default 100) evaluations, each graph captured once per (plan, length) from the same pre-bound C-ABI calls and launched
once before any timed region.  W and K are exact (a remainder gets a graph of its own).  `--graph 0` = the eager loop, one call per step
(the default for a step of more than one launch: c4 / c5 with `--two-launch` measured slower captured).

Prints ONE JSON line (rank 0).  `roofline` describes the step's dominant kernel (the fused rollout): `achieved` = its algorithmic
bytes/sample x samples per launch / its average launch duration (HIP events on the launch stream: around the timed region when
the step is that one launch, around a loop of that kernel alone otherwise); `roofline.step` prices the whole step.
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
# per-launch counters recorded by tools/run_r06_profiles.sh (earlier rounds' files as fall-backs)
PMC_FILES = ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_hbm_traffic.json")
DEFAULT_SHAPE = {"c2": (4096, 64), "c3": (4096, 64), "c4": (4096, 64), "c5": (2048, 128)}     # per-GPU batch x horizon
SIGMA_GP, T_GP = 0.1, 5.0        # config 5's GP prior: sigma_gp of the reference's planner parameters (env_spheres_3d.py:57), 5 s trajectories


def algorithmic_bytes_per_sample(D, L, n_grid_links=0, esz=4):
    # read q (D) + write link positions (3L) + gradient (D), `esz` bytes each, + cost (4); the voxel-grid scene
    # adds one 16-byte gather (sdf + stored gradient of the nearest cell) per collision link
    return esz * D + 3 * esz * L + 4 + esz * D + 16 * n_grid_links


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--scene", default="spheres", choices=["spheres", "grid", "shelf", "maze"])
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default: 4096; c5: 2048)")
    ap.add_argument("--horizon", type=int, default=None, help="time steps per trajectory (default: 64; c5: 128)")
    ap.add_argument("--reduce-every", type=int, default=64,
                    help="the planner's exchange cadence; a short run uses min(this, steps) so that its timed region contains a collective")
    ap.add_argument("--graph", type=int, default=None,
                    help="launch mode: the steps are captured into hipGraphs of at most this many evaluations and replayed; 0 = eager "
                         "launches, one pre-bound C-ABI call per step.  Default: 100 for one-launch steps, eager for c4's two-kernel step")
    ap.add_argument("--weights", default=None, help="experiment: w_self,w_obj,w_ws,w_ee override (reported in config)")
    ap.add_argument("--no-pos", action="store_true", help="experiment: do not write link positions")
    ap.add_argument("--q", default="iid", choices=["iid", "smooth"],
                    help="c2 / c3 input: independent uniform configurations (default), or trajectories that are smooth along the horizon "
                         "(a random walk of 0.02 rad steps: what a planner feeds; matters for the voxel-grid scene's gathers)")
    ap.add_argument("--independent-streams", type=int, default=0,
                    help="also report the throughput of unrelated batches alternated over this many HIP streams (secondary "
                         "figure `independent_batches`; off by default so that a rocprofv3 run of the default command sees only "
                         "back-to-back launches of one stream)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--two-launch", action="store_true", help="c5: the round-3 form (fp16 rollout, then the GP prior accumulated) instead of the fused launch; "
                                                              "c4: the fused rollout, then the Jacobian kernel, instead of the one-launch form")
    ap.add_argument("--no-out-of-cache", action="store_true", help="skip the secondary beyond-the-Infinity-Cache measurement (c2, N = 1)")
    ap.add_argument("--dist-backend", default="nccl", help="debug: 'gloo' + --single-device lets the N>1 control flow run on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--same-q", action="store_true", help="debug: every rank draws the same q (all-reduced sums == N x rank 0's)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "rccl"],
                    help="how the packed sums travel: 'p2p' = the peer-to-peer mailbox (trk_mailbox_*: every rank stores its row into every "
                         "peer's device memory over xGMI and adds the rows in rank order; one kernel, captured into the step graphs), "
                         "'rccl' = an all-reduce on a side stream, 'auto' (default) = p2p when its known-answer validation passes on "
                         "every rank, else rccl")
    ap.add_argument("--native-rccl", action="store_true",
                    help="issue the planner's exchange straight on librccl (torch_robotics_amd.distributed.RcclAllReduce: one ctypes "
                         "call per collective) instead of torch.distributed.all_reduce; the barriers stay on torch.distributed")
    ap.add_argument("--side-priority", type=int, default=0, help="experiment: HIP priority of the side stream the exchanges run on (-1 = high)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: run the N > 1 code path (process group, side-stream all-reduce, in-stream barriers, multi_gpu section) "
                         "even with ONE rank -- the only way to put the RCCL calls on hardware on a 1-GPU box")
    args = ap.parse_args(argv)
    b, h = DEFAULT_SHAPE[args.config]
    args.batch = b if args.batch is None else args.batch
    args.horizon = h if args.horizon is None else args.horizon
    return args


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


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of /proc/cpuinfo among the CPUs of the affinity mask
    (SMT siblings count once -- what the OpenMP runtime picks by default on the N = 1 run); the mask's size when that cannot be read."""
    allowed = os.sched_getaffinity(0)
    try:
        cores, cpu, pkg = set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu, pkg = int(v), None
            elif k == "physical id":
                pkg = int(v)
            elif k == "core id" and cpu in allowed:
                cores.add((pkg, int(v)))
        return len(cores) or len(allowed)
    except Exception:
        return len(allowed)


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


W_C2, W_C3 = (0.0, 1.0, 0.0, 1.0), (1.0, 1.0, 1.0, 1.0)


class Workload:
    """One BASELINE config on one rank: the pre-bound launches of a step, what the exchange packs, and the oracle's side of it."""
    extras = ()             # [(label, launch(stream), algorithmic bytes per sample)]: launches of a step after the fused rollout
    traj_cost = None        # per-trajectory cost evaluated next to the rollout (GP prior): joins packed[0]
    plan3 = None            # c2 / c3: configs[2]'s objective stack on the same shard
    n_grid_links = 0
    esz = 4
    extra_bps = 0           # bytes per sample the dominant kernel moves beyond q, positions, cost, gradient (c5 fused: qd in, gqd out)

    def step(self, bs_ptr, s):
        self.plan.launch(bs_ptr, s)
        for _, fn, _ in self.extras:
            fn(s)

    @property
    def bps_step(self):
        return self.bps + sum(b for _, _, b in self.extras)


def smooth_trajectories(torch, B, H, D, gen, dev, lo, hi):
    """random walks of 0.02 rad steps around a uniform start, clipped to the joint limits"""
    start = lo + (hi - lo) * torch.rand(B, 1, D, generator=gen, device=dev)
    walk = torch.cumsum(torch.randn(B, H, D, generator=gen, device=dev) * 0.02, 1)
    return torch.minimum(torch.maximum(start + walk, lo), hi).contiguous()


def build_workload(args, tra, ops, torch, dev, rank):
    import numpy as np
    ta = dict(device=dev, dtype=torch.float32)
    B, H = args.batch, args.horizon
    gen = torch.Generator(device=dev).manual_seed(1234 + (0 if args.same_q else rank))
    wl = Workload()
    wl.B, wl.H, wl.cfg, wl.gen = B, H, args.config, gen
    if args.config in ("c2", "c3"):
        robot, task, scene_text = make_task(tra, args.scene, ta)
        weights = W_C2 if args.config == "c2" else W_C3
        if args.weights:
            weights = tuple(float(v) for v in args.weights.split(","))
        kin = robot.diff_panda._kin
        D, L = robot.q_dim, kin.n_links
        if args.q == "smooth":
            q = smooth_trajectories(torch, B, H, D, gen, dev, robot.q_min.to(dev), robot.q_max.to(dev))
        else:
            q = robot.random_q(B * H, generator=gen).reshape(B, H, D).contiguous()
        model, cm = task._fused_handles(dev)
        wl.plan = ops.RolloutPlan(model, cm, weights, q, want_pos=not args.no_pos)
        wl.plan3 = wl.plan if weights == W_C3 else ops.RolloutPlan(model, cm, W_C3, q, want_pos=not args.no_pos)
        wl.spec_fn = task.build_cost_spec
        wl.n_grid_links = len(cm.spec.obj_link_idx) if (cm.spec.grid is not None and weights[1] != 0.0) else 0
        objectives = {"c2": "SDF-obstacle + EE-tracking", "c3": "self-collision + SDF-obstacle + workspace box + EE-tracking"}[args.config]
        wl.metric = "FK+cost+grad rollouts/sec (batch x horizon), Panda 7-DOF"
        wl.text = (f"BASELINE configs[{1 if args.config == 'c2' else 2}]{'' if args.config == 'c2' else ' objective stack'}: Franka Panda "
                   f"({L} links, {D} DOF), batch={B} x horizon={H} per GPU, fused FK + {objectives} cost + gradient ({scene_text}), "
                   f"q resident in HBM" + (", smooth trajectories" if args.q == "smooth" else ""))
        wl.dtype, wl.random_q = "f32", lambda n: robot.random_q(n, generator=gen)
        wl.make_plan = lambda qq: ops.RolloutPlan(model, cm, weights, qq, want_pos=not args.no_pos)
    else:
        from torch_robotics_amd import codegen
        from torch_robotics_amd.costmodel import CostModelSpec
        ident = "ur10_allegro" if args.config == "c4" else "dual_panda"
        kin, tmpl = codegen.template_for(ident)
        env = tra.EnvSpheres3D(tensor_args=ta)
        spec = CostModelSpec(n_links_in=kin.n_links)
        spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
        spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
        spec.objects = [o.as_object() for o in env.obj_fixed_list]
        spec.ee_link = tmpl.ee_link
        Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
        if tmpl.ee2_link >= 0:          # two-arm template: the second arm tracks its own target
            spec.ee2_link = tmpl.ee2_link
            Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
        spec.validate()
        model, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
        D, L = kin.n_dofs, kin.n_links
        weights = W_C2 if not args.weights else tuple(float(v) for v in args.weights.split(","))
        wl.spec_fn = lambda: spec
        scene_text = "EnvSpheres3D, 10 spheres, analytic SDF"
        if args.config == "c4":
            q = ((torch.rand(B, H, D, generator=gen, **ta) - 0.5) * 3.0).contiguous()
            ee = int(kin.name_to_idx["ee_link"])
            wl.ee = ee
            if args.two_launch or args.no_pos:
                # the round-2 .. 5 form: the fused rollout, then the Jacobian kernel (a second walk of the chain, q read twice)
                wl.plan = ops.RolloutPlan(model, cm, weights, q, want_pos=not args.no_pos)
                jac = ops.JacobianPlan(model, q.reshape(B * H, D), ee)
                wl.jac = jac
                wl.extras = [("geometric Jacobian of ee_link (trk_fk_jacobian: pos, quat, lin_jac, ang_jac)", jac.launch, 4 * D + 28 + 24 * D)]
                how = "then the geometric Jacobian of ee_link (two launches)"
            else:
                # ONE launch (trk_rollout_jacobian_cost_grad, round 6): the Jacobian's columns are read out of the poses the rollout holds
                wl.plan = ops.RolloutJacobianPlan(model, cm, weights, q, ee)
                wl.jac = wl.plan
                wl.extra_bps = 28 + 24 * D                                # + pos, quat, lin_jac, ang_jac out (q is read once)
                how = "and the geometric Jacobian of ee_link in the SAME launch"
            wl.metric = "FK+Jacobian+cost rollouts/sec (batch x horizon), UR10+Allegro 22-DOF"
            wl.text = (f"BASELINE configs[3]: UR10 + Allegro hand ({L} links, {D} DOF), batch={B} x horizon={H} per GPU, fused FK + "
                       f"SDF-obstacle + EE-tracking cost + gradient + link positions ({scene_text}), {how}; "
                       f"q resident in HBM")
            wl.dtype = "f32"
        else:
            dt = T_GP / H
            q = torch.cumsum(torch.randn(B, H, D, generator=gen, **ta) * 0.02, 1) + (torch.rand(B, 1, D, generator=gen, **ta) - 0.5) * 2.0
            qd = torch.zeros_like(q)
            qd[:, :-1] = (q[:, 1:] - q[:, :-1]) / dt                     # the velocities of the same trajectories
            qd[:, -1] = qd[:, -2]
            qh, qdh = q.half().contiguous(), qd.half().contiguous()
            # loss scale: worst case of the GP gradient over trajectories bounded like these, + the collision / EE gradient's size
            gs = ops.gp_grad_scale(dt, SIGMA_GP, 1.0, float(qh.abs().max()), float(qdh.abs().max()), extra=64.0)
            wl.qd, wl.dt, wl.grad_scale = qdh, dt, gs
            if args.two_launch:
                # the round-3 form: the fp16 rollout, then the prior accumulated into its gradient (a read-modify-write of gq)
                wl.plan = ops.RolloutPlan(model, cm, weights, qh, want_pos=not args.no_pos, grad_scale=gs)
                wl.gqd = torch.zeros_like(qh)
                wl.gp = ops.GPPriorPlan(qh, qdh, dt, SIGMA_GP, 1.0, accumulate_into=(wl.plan.gq, wl.gqd), grad_scale=gs)
                wl.traj_cost = wl.gp.cost
                # GP prior: read q, qd (2 x 2D), read-modify-write gq (2 x 2D), write gqd (2D), cost per trajectory
                wl.extras = [("GP prior cost + gradient accumulated into gq / gqd (trk_gp_prior_cost_grad, fp16 I/O)", wl.gp.launch, 2 * 5 * D)]
                how = "fused FK + SDF-obstacle + EE-tracking on both arms + gradient, then the GP prior accumulated into the same gradient"
            else:
                # ONE launch (trk_rollout_gp_cost_grad): q, qd read once, gq, gqd written once
                wl.plan = ops.RolloutGpPlan(model, cm, weights, qh, qdh, dt, SIGMA_GP, 1.0, want_pos=not args.no_pos, grad_scale=gs)
                wl.gqd = wl.plan.gqd
                wl.extra_bps = 2 * 2 * D                                 # + qd in, gqd out
                how = "ONE launch: fused FK + SDF-obstacle + EE-tracking on both arms + the GP prior + both gradients"
            wl.metric = "FK+cost+grad rollouts/sec (batch x horizon), dual Panda 14-DOF, fp16 I/O"
            wl.text = (f"BASELINE configs[4]: dual Panda ({L} links, {D} DOF), batch={B} x horizon={H} per GPU (8192 x 128 over 4 GPUs), "
                       f"fp16 q / qd / link positions / gradients in HBM (loss scale 2^{int(np.log2(gs))}), fp32 arithmetic and cost: {how} "
                       f"(sigma_gp = {SIGMA_GP}, dt = {T_GP}/{H}); q, qd resident in HBM")
            wl.dtype, wl.esz, q = "f16", 2, qh
        wl.random_q = None
    wl.q, wl.model, wl.cm, wl.kin, wl.weights, wl.D, wl.L = q, model, cm, kin, weights, D, L
    wl.bps = algorithmic_bytes_per_sample(D, L if not args.no_pos else 0, wl.n_grid_links, wl.esz) + wl.extra_bps
    return wl


def cpu_baseline(wl, args, torch, seconds):
    """The C oracle on a bounded sample of rank 0's batch: all host threads (`value`) and one thread."""
    import numpy as np
    from oracle import oracle as orc          # checker / baseline only: never on the product path
    spec = wl.spec_fn()
    if spec.grid is not None:                 # the oracle reads host arrays
        spec.grid = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in spec.grid.items()}
    o = orc.Oracle(wl.kin, spec)
    D, H = wl.D, wl.H
    q_host = wl.q.reshape(-1, D).float().cpu().numpy()           # c5: the fp16-rounded trajectories, widened
    qd_host = wl.qd.reshape(-1, D).float().cpu().numpy() if wl.cfg == "c5" else None
    cores = orc.max_threads()
    unit = H if wl.cfg == "c5" else 1                            # c5 evaluates whole trajectories (the GP prior couples time steps)

    def one(n):
        o.rollout(q_host[:n], wl.weights, "f32")
        if wl.cfg == "c4":
            o.jacobian(q_host[:n], None, wl.ee, "f32")
        elif wl.cfg == "c5":
            orc.gp_prior(q_host[:n].reshape(-1, H, D), qd_host[:n].reshape(-1, H, D), wl.dt, SIGMA_GP, 1.0, "f32")

    def timed(n, reps):
        best = 1e30
        for _ in range(reps):
            t = time.perf_counter(); one(n); best = min(best, time.perf_counter() - t)
        return best

    def whole(n):
        return max(unit, int(n) // unit * unit)
    # all threads: ~2/3 of the budget; one thread: ~1/3 (both bounded samples of rank 0's batch)
    probe = whole(min(16384, len(q_host)))
    dt_ = timed(probe, 1)
    n_all = whole(min(len(q_host), max(probe, probe / dt_ * seconds * 2 / 9)))
    t_all = timed(n_all, 3)
    orc.set_threads(1)
    p1 = whole(2048)
    dt1 = timed(p1, 1)
    n_one = whole(min(len(q_host), max(p1, p1 / dt1 * seconds / 6)))
    t_one = timed(n_one, 2)
    orc.set_threads(cores)
    what = {"c2": "fused rollout", "c3": "fused rollout", "c4": "rollout + geometric Jacobian of ee_link",
            "c5": "rollout (fp32 on the fp16-rounded q) + GP prior"}[wl.cfg]
    # BASELINE.md section 2 (measured once, in the survey container, by importing the reference itself; it cannot travel to this box):
    # labelled so that the C restatement above is not mistaken for the (much slower) reference
    survey = {"c2": {"value": 8.3e4, "what": "Task-API path (fk_map_collision + EE pose: 2 x FK), object + EE cost + backward, N = 262 144: 3.17 s",
                     "single_fk": {"value": 1.53e5, "what": "1 x FK + object-SDF + EE cost + backward, N = 32 768: 214 ms"}},
              "c3": {"value": 5.5e4, "what": "Task-API path, self + object + workspace box + EE + backward, N = 262 144: 4.79 s",
                     "single_fk": {"value": 1.36e5, "what": "1 x FK + full stack + backward, N = 32 768: 242 ms"}}}.get(wl.cfg)
    ref_fig = None if survey is None else {
        "label": "reference PyTorch-CPU, 8 vCPU (Intel Xeon 2.1 GHz, 8 torch threads), survey container -- NOT measured on this box",
        "unit": "rollouts/s", "source": "BASELINE.md section 2", **survey}
    return {"value": n_all / t_all, "unit": "rollouts/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
            "reference_pytorch_cpu_survey": ref_fig,
            "sample": f"first {n_all} of the {wl.B * H} samples of rank 0's batch, C oracle ({what}; fp32, OpenMP over samples, "
                      f"{cores} threads), best of 3",
            "one_core": {"value": n_one / t_one, "unit": "rollouts/s", "cores": 1,
                         "sample": f"first {n_one} samples, same code on 1 thread, best of 2"}}


def exchange_pieces(count, cadence):
    """The exchange schedule of `count` steps: [(steps, exchange afterwards?)] -- one exchange per `cadence` steps, fired in the middle of
    its interval (after step j whenever j % cadence == cadence // 2); cadence 0 = no exchange."""
    if not cadence:
        return [(count, False)] if count else []
    out_, last = [], 0
    for j in range(1, count + 1):
        if j % cadence == cadence // 2:
            out_.append((j - last, True)); last = j
    if last < count:
        out_.append((count - last, False))
    return out_


def exchange_schedule(count, cadence, graph_steps, mailbox, n_slots):
    """What one call of run() issues for `count` steps: a list of ("graph", segs) / ("steps", n) / ("exchange",) items.
    graph_steps: evaluations per captured graph at most (0 = eager launches).  segs = (a0, a1, ..., ak): a0 steps, exchange, a1 steps,
    exchange, ..., ak steps.  With the mailbox and capture on, the exchanges ride INSIDE the graphs (<= graph_steps steps and <= n_slots
    exchanges each: the driver's 20 timed steps with their exchange are ONE graph launch; an exchange that falls on a graph boundary
    opens the next graph -- its pack reads the previous graph's last step); otherwise plain-step graphs (or eager steps) with an eager
    exchange between them.  Pure function: tests/test_bench_launch.py checks step and exchange counts for many (count, cadence, size)."""
    items = []
    S = int(graph_steps)
    if S and mailbox and cadence:
        segs, steps_in = [0], 0
        for n, ex in exchange_pieces(count, cadence):
            while n > 0:
                take = min(n, S - steps_in)
                if take == 0:                                # the graph is full: close it
                    items.append(("graph", tuple(segs))); segs, steps_in = [0], 0
                    continue
                segs[-1] += take; steps_in += take; n -= take
            if ex:
                if len(segs) > n_slots or steps_in >= S:    # start a new graph; the exchange opens it
                    items.append(("graph", tuple(segs))); segs, steps_in = [0], 0
                segs.append(0)
        if steps_in or len(segs) > 1:
            items.append(("graph", tuple(segs)))
        return items
    for n, ex in exchange_pieces(count, cadence):
        while n > 0 and S:
            items.append(("graph", (min(n, S),))); n -= min(n, S)
        if n:
            items.append(("steps", n))
        if ex:
            items.append(("exchange",))
    return items


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
    wl = build_workload(args, tra, ops, torch, dev, rank)
    B, H, D, L = wl.B, wl.H, wl.D, wl.L
    plan, model = wl.plan, wl.model
    # the fused kernel leaves one partial cost sum per wavefront (= per trajectory at horizon 64); a rank folds them
    # into a scalar with the deterministic reduce kernel only when a collective needs it
    nb = ops.n_blocks(B * H)
    block_sums = torch.zeros(nb, **ta)
    bs_ptr = block_sums.data_ptr()
    # The planner's cadence: one exchange per `reduce_every` evaluations, fired in the middle of its interval (see run()).  A run shorter
    # than an interval shrinks the interval to the run, so that the driver's 20 timed steps contain an exchange and `value` is never a
    # kernel-only figure (`multi_gpu.collectives_in_timed_region` counts them).
    R = max(1, min(args.reduce_every, args.steps))
    # what a sharded planner exchanges: one packed fp32 buffer [sum cost | sum_b cost(h) | sum_b grad(h, d)]
    n_slots = 128
    packed = torch.zeros((n_slots, 1 + H + H * D), **ta)
    stream = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev, priority=args.side_priority) if distributed else None

    packers = {}

    def pack_sums(pl, buf, s=None):
        """The sums of the latest evaluation of plan `pl` -> buf: one launch of trk_pack_sums on the launch stream."""
        pk = packers.get(id(pl))
        if pk is None:
            pk = packers[id(pl)] = ops.PackedSums(pl, block_sums, wl.traj_cost if pl is plan else None)
        pk.pack(buf, stream.cuda_stream if s is None else s)

    native = None
    if distributed and args.native_rccl:
        from torch_robotics_amd.distributed import RcclAllReduce
        native = RcclAllReduce(dev)

    # The peer-to-peer mailbox (SURVEY 8e's alternative; csrc/trk_exchange.hip).  Every rank constructs it (the handles travel through
    # the process group); a known-answer validation on all ranks decides whether it is used -- a box where peer stores do not arrive
    # times out inside the kernel (no hang) and the run keeps the all-reduce.
    mailbox, mailbox_note = None, None
    if distributed and args.exchange in ("auto", "p2p"):
        from torch_robotics_amd.distributed import MailboxAllReduce
        try:
            mailbox = MailboxAllReduce(dev, 1 + H + H * D, n_slots=8)
            if not mailbox.validate():
                mailbox_note = "validation failed (a peer's stores did not arrive / wrong sums): " + repr(mailbox.status())
                mailbox.close(); mailbox = None
        except Exception as e:                                   # e.g. hipIpcOpenMemHandle refused
            mailbox_note, mailbox = f"{type(e).__name__}: {e}", None
        if mailbox is None:
            if args.exchange == "p2p":
                raise SystemExit(f"--exchange p2p: the mailbox is not usable here: {mailbox_note}")
            if rank == 0:
                print(f"[bench] peer-to-peer mailbox not usable ({mailbox_note}); the exchange stays an all-reduce", file=sys.stderr)
    reduced = torch.zeros((n_slots, 1 + H + H * D), **ta) if mailbox is not None else None      # the mailbox's results (ring, like `packed`)

    slot_free = [None] * n_slots        # per exchange buffer: the side-stream event after which it may be packed again

    mb_pending = [None]                 # eager mailbox mode: the ring slot whose rows have been sent but not yet received

    def mailbox_flush():
        if mb_pending[0] is not None:
            mailbox.recv(reduced[mb_pending[0]], stream.cuda_stream)
            mb_pending[0] = None

    def reduce_slot(pl, k):
        # sums of the latest evaluation -> one small all-reduce, off the launch stream.  The buffers are a ring: before a slot is
        # packed again the launch stream waits for the collective that last used it (a planner's bounded look-ahead) -- with one
        # exchange per step (`multi_gpu.every_step`) that is what ties the launch rate to the exchange rate.
        buf = packed[k]
        if mailbox is not None:
            # the mailbox needs no second stream: SEND (stores + flag, never waits) right behind the pack; the RECEIVE of this exchange
            # is issued later on the same stream -- before the next exchange or at the end of run() -- when the peers' rows have arrived
            mailbox_flush()
            pack_sums(pl, buf)
            mailbox.send(buf, stream.cuda_stream)
            mb_pending[0] = k
            return
        if slot_free[k] is not None:
            stream.wait_event(slot_free[k])
        pack_sums(pl, buf)
        ev = torch.cuda.Event()
        ev.record(stream)
        side.wait_event(ev)
        if native is not None:
            native.all_reduce_sum_(buf, side.cuda_stream)
        else:
            with torch.cuda.stream(side):
                dist.all_reduce(buf)
        done = torch.cuda.Event()
        done.record(side)
        slot_free[k] = done

    def step_of(pl):
        return wl.step if pl is plan else pl.launch

    # Launch mode.  The inner loop is launch-bound bookkeeping around a 9 us kernel, so by default it is captured: every run of
    # consecutive evaluations between two exchanges becomes hipGraph replays of at most `--graph` evaluations each (the same pre-bound
    # C-ABI calls, captured once per (plan, length) and replayed).  Same-box, 4096 x 64 Panda: 2000 steps 9.39 - 9.43 us eager, 9.08 us as
    # 100-evaluation graphs; the driver's 20 steps 10.9 - 11.7 us eager, 10.5 us as one graph; c5 21.6 -> 21.0 us; c3 unchanged.  c4's step
    # alternates two different kernels and measured SLOWER captured (60.6 -> 63.7 us): it stays eager unless asked (profiles/
    # r04_graph_vs_eager.txt).  `--graph 0` keeps the eager loop everywhere.
    # The gloo debug mode (ranks sharing one GPU, exchanges through host copies) also stays eager: a graph launch between synchronous
    # copies from two processes costs milliseconds there (every_step 0.37 -> 6.2 ms).
    S = max(0, args.graph) if args.graph is not None else (0 if (wl.extras or (distributed and args.dist_backend == "gloo")) else 100)
    graphs = {}
    S_live = [S]                 # becomes 0 if a capture fails
    gstream = torch.cuda.Stream(dev) if S else None

    def graph_for(pl, segs):
        """The graph of `segs` = (a0, a1, ..., ak): a0 steps of plan pl, exchange, a1 steps, exchange, ..., ak steps (k exchanges; an int
        = that many plain steps).  An exchange inside a graph is the mailbox form, all on ONE stream: trk_pack_sums + trk_mailbox_send
        behind the step whose sums travel, trk_mailbox_recv behind the steps of the next segment -- by then the peers' rows are there,
        the wait costs nothing.  (A forked branch for the exchange was measured first: HIP serialised the branch in front of the
        following steps anyway and the forked graph's launches came in bursts with 4 - 9 us gaps, profiles/r05_exchange_trace_c2_fork.txt.)
        Captured once, launched once; None when capture is off / failed."""
        if not S_live[0]:
            return None
        segs = (segs,) if isinstance(segs, int) else tuple(segs)
        g = graphs.get((id(pl), segs))
        if g is None:
            fn = step_of(pl)
            try:
                with torch.cuda.stream(gstream):
                    for i in range(3):                          # nothing lazy may happen inside the capture
                        fn(bs_ptr, gstream.cuda_stream)
                    if len(segs) > 1:
                        pack_sums(pl, packed[0], gstream.cuda_stream)
                gstream.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: the process group's watchdog thread may query events while this thread captures
                with torch.cuda.graph(g, stream=gstream, capture_error_mode="thread_local"):
                    cur = torch.cuda.current_stream(dev).cuda_stream
                    for j, n in enumerate(segs):
                        if j > 0:                               # exchange j - 1: pack + send here, its receive behind the NEXT segment
                            k = (j - 1) % n_slots
                            pack_sums(pl, packed[k], cur)
                            mailbox.send(packed[k], cur)
                        for i in range(n):
                            fn(bs_ptr, cur)
                        if j > 0:
                            mailbox.recv(reduced[(j - 1) % n_slots], cur)
                g.replay()                                      # the first launch of an instantiated graph uploads it
                torch.cuda.synchronize(dev)
            except Exception as e:                              # a box whose runtime refuses the capture still measures: eager loop
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
                S_live[0] = 0
                torch.cuda.synchronize(dev)
                return None
            graphs[(id(pl), segs)] = g
        return g

    def pieces(count, cadence):
        return exchange_pieces(count, cadence)

    def schedule(count, cadence):
        return exchange_schedule(count, cadence, S if S_live[0] else 0, mailbox is not None, n_slots)

    def prepare(pl, count, cadence):
        """capture (and launch once) every graph run(pl, count, cadence) will replay -- outside any timed region"""
        for it in schedule(count, cadence):
            if it[0] == "graph" and S_live[0]:
                graph_for(pl, it[1])

    slot = [0]

    def run(pl, count, cadence):
        """`count` steps; cadence > 0: one exchange per `cadence` steps of this call, fired in the MIDDLE of its interval.  Mailbox mode:
        the exchange is part of the captured interval (no host work at all).  All-reduce mode: the host is then several launches ahead
        of the GPU (handing a collective to torch.distributed costs the launch thread ~60 us) and half an interval is left for the
        collective to complete."""
        s = stream.cuda_stream
        fn = step_of(pl)
        for it in schedule(count, cadence):
            if it[0] == "graph":
                g = graph_for(pl, it[1])
                if g is not None:
                    g.replay()
                    slot[0] += len(it[1]) - 1
                    continue
                for j, n in enumerate(it[1]):                   # a failed capture: the same items eagerly
                    if j > 0:
                        reduce_slot(pl, slot[0] % n_slots); slot[0] += 1
                    for _ in range(n):
                        fn(bs_ptr, s)
            elif it[0] == "steps":
                for _ in range(it[1]):
                    fn(bs_ptr, s)
            else:
                reduce_slot(pl, slot[0] % n_slots)
                slot[0] += 1
        if mailbox is not None:
            mailbox_flush()

    flag = torch.zeros(1, **ta)

    def barrier_in_stream():
        """A barrier that costs no host round trip: a one-element all-reduce enqueued on the launch stream completes on a rank
        only after every rank has reached it, so the `torch.cuda.synchronize()` that follows returns when all ranks are done."""
        if native is not None:
            # two communicators on one device (torch's and RcclAllReduce's own) must not run collectives concurrently: the barrier
            # goes behind whatever the side stream still has in flight
            stream.wait_stream(side)
        dist.all_reduce(flag)

    n_coll = [0]            # collectives inside the latest timed region
    per_rank = [None]       # the latest timed region: every rank's wall clock and launch-stream time

    def measure(pl, cadence, steps=None, warmup=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize brackets; returns (wall s, event ms),
        the wall time already as the maximum over the ranks."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        prepare(pl, warmup, cadence); prepare(pl, steps, cadence)
        run(pl, warmup, cadence)
        if distributed:
            barrier_in_stream()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        slots_before = slot[0]
        run(pl, steps, cadence)
        n_coll[0] = slot[0] - slots_before
        ev1.record(stream)
        ev_side = None
        if side is not None and cadence:
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
        ev_ms = ev0.elapsed_time(ev1)
        if distributed:
            # every rank's own figures (wall clock, launch stream by events): the MAX is the result, the spread says who was late
            mine = torch.tensor([wall, ev_ms * 1e-3], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank[0] = {"wall_us": [round(float(t[0]) * 1e6, 1) for t in allr], "launch_stream_us": [round(float(t[1]) * 1e6, 1) for t in allr]}
            wall = max(float(t[0]) for t in allr)
        return wall, ev_ms

    # one rehearsal of the whole measurement (discarded): the first pass through the event / sync / launch code paths of a
    # fresh process costs ~10 us more, which matters when the driver asks for only 20 timed steps (~200 us of GPU work)
    cadence = R if distributed else 0

    def mailbox_healthy():
        """Collective: False on EVERY rank when ANY rank's mailbox kernels gave up waiting (an in-kernel time-out costs seconds and leaves
        an incomplete sum) -- the run then drops the mailbox, all ranks together, and measures with the all-reduce instead of reporting a
        polluted figure or dying."""
        bad = torch.tensor([float(mailbox.status()[1] > 0)], device="cpu" if dist.get_backend() == "gloo" else dev)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        return float(bad.item()) == 0.0

    measure(plan, cadence)
    elapsed, ev_ms = measure(plan, cadence)
    if mailbox is not None and not mailbox_healthy():
        mailbox_note = "dropped after the first measurement: exchanges timed out (" + repr(mailbox.status()) + ")"
        if rank == 0:
            print(f"[bench] peer-to-peer mailbox {mailbox_note}; measuring again with the all-reduce", file=sys.stderr)
        torch.cuda.synchronize(dev)
        graphs.clear()
        mailbox.close(); mailbox = None
        measure(plan, cadence)
        elapsed, ev_ms = measure(plan, cadence)
    n_coll_value = n_coll[0]
    per_rank_value = per_rank[0]

    samples_per_step = B * H * world
    value = samples_per_step * args.steps / elapsed
    bps = wl.bps
    bytes_per_launch = bps * B * H
    step_s = ev_ms * 1e-3 / args.steps
    if wl.extras:
        # the step has more launches than the dominant kernel: time that kernel alone, same stream, same brackets
        def only_rollout(n):
            for _ in range(n):
                plan.launch(bs_ptr, stream.cuda_stream)
        only_rollout(max(3, min(args.warmup, 50)))
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); only_rollout(args.steps); e1.record(stream)
        torch.cuda.synchronize(dev)
        launch_s = e0.elapsed_time(e1) * 1e-3 / args.steps
        wl.step(bs_ptr, stream.cuda_stream)                     # leave the buffers as a whole step leaves them
        torch.cuda.synchronize(dev)
    else:
        launch_s = step_s
    achieved = bytes_per_launch / launch_s / 1e9

    # sanity: the outputs of the last step are finite and the cost sums agree with the per-sample costs
    assert torch.isfinite(plan.cost).all() and torch.isfinite(plan.gq.float()).all()
    if plan.gq.dtype == torch.float16:          # nothing saturated either: the loss scale holds
        assert float(plan.gq.float().abs().max()) < 65504.0 and float(wl.gqd.float().abs().max()) < 65504.0
    last = ops.reduce_sum(block_sums).item()
    ref = plan.cost.double().sum().item()
    assert abs(last - ref) <= 1e-4 * abs(ref) + 1e-3, (last, ref)

    # PMC figures of this workload recorded under profiles/ (separate rocprofv3 --pmc passes of this very command)
    kind = "specialized" if model.specialized else "table-driven"
    pmc_key = f"{args.config}:{args.scene}:{B}x{H}:{kind}" + (":smooth" if args.q == "smooth" else "")
    pmc, pmc_src = (None, None) if (args.weights or args.no_pos) else pmc_record(pmc_key)
    if pmc is None and args.scene == "spheres" and args.q == "iid" and not (args.weights or args.no_pos):
        pmc, pmc_src = pmc_record(f"{args.config}:{B}x{H}:{kind}")                  # round-2 key format
    traffic = pmc.get("traffic_bytes_per_launch") if pmc else None
    valu_per_wave = pmc.get("valu_insts_per_wave") if pmc else None
    valu_frac = None if valu_per_wave is None else valu_per_wave * nb / launch_s / VALU_PEAK_WAVE_INSTS_PER_S

    out = {
        "metric": wl.metric,
        "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": wl.dtype, "data": "synthetic",
        "config": {"workload": wl.text,
                   "global_batch": B * world, "horizon": H, "parallelism": f"batch-sharded x{world}",
                   "scene": args.scene, "objectives": args.config,
                   "launch": ("hipGraph replays of <= %d captured evaluations (pre-bound C-ABI calls)" % S) if S_live[0] else "eager, pre-bound C-ABI call",
                   "launches_per_step": 1 + len(wl.extras),
                   "kernel": kind,
                   "reduce_every": args.reduce_every if distributed else None,
                   "reduce_every_effective": R if distributed else None,
                   **({"q": "smooth"} if args.q == "smooth" else {}),
                   **({"grad_scale": wl.grad_scale} if args.config == "c5" else {}),
                   **({"experiment_weights": list(wl.weights)} if args.weights else {}),
                   **({"experiment_no_pos": True} if args.no_pos else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": pmc_src if traffic else None,
                     "kernel": ("fused rollout + GP prior (trk_rollout_gp_cost_grad)" if isinstance(plan, ops.RolloutGpPlan) else
                                "fused rollout + geometric Jacobian (trk_rollout_jacobian_cost_grad)" if isinstance(plan, ops.RolloutJacobianPlan) else
                                "fused rollout (trk_rollout_cost_grad%s)" % ("_f16" if wl.esz == 2 else "")),
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
    if wl.extras:
        out["roofline"]["step"] = {"bytes_per_sample": wl.bps_step, "step_us": step_s * 1e6,
                                   "achieved": wl.bps_step * B * H / step_s / 1e9, "frac": wl.bps_step * B * H / step_s / 1e9 / HBM_PEAK_GBS,
                                   "launches": ["fused rollout"] + [lbl for lbl, _, _ in wl.extras]}

    if (rank == 0 and not distributed and args.config == "c2" and not args.no_out_of_cache and wl.random_q is not None
            and bytes_per_launch < 256e6):
        # Secondary figure: the same kernel on a batch whose working set exceeds the 256 MB Infinity Cache (8 x the trajectories:
        # 403 MB per launch at the default size), i.e. what the DRAM itself sustains for this read / write mix.
        B_big = 8 * B
        q_big = wl.random_q(B_big * H).reshape(B_big, H, D).contiguous()
        big = wl.make_plan(q_big)
        sums_big = torch.zeros(ops.n_blocks(B_big * H), **ta)
        n_big = max(20, min(200, args.steps // 8))
        for _ in range(10):
            big.launch(sums_big.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        g_big = None
        if S_live[0]:                                            # the headline's launch mode: a captured run of launches, replayed
            try:
                g_big = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_big, stream=gstream, capture_error_mode="thread_local"):
                    for _ in range(n_big):
                        big.launch(sums_big.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
                g_big.replay()
                torch.cuda.synchronize(dev)
            except Exception:
                g_big = None
                torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        if g_big is not None:
            g_big.replay()
        else:
            for _ in range(n_big):
                big.launch(sums_big.data_ptr(), stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        big_s = e0.elapsed_time(e1) * 1e-3 / n_big
        out["roofline"]["frac_out_of_cache"] = bps * B_big * H / big_s / 1e9 / HBM_PEAK_GBS
        out["roofline"]["out_of_cache"] = {"batch": B_big, "working_set_MB": round(bps * B_big * H / 1e6, 1), "launch_us": big_s * 1e6,
                                           "steps": n_big, "launch": "hipGraph replay" if g_big is not None else "eager",
                                           "stores": "non-temporal (the launch chose the F32Stream instantiation: working set > 256 MiB)"
                                                     if bps * B_big * H > 256 * 1024 * 1024 else "write-through"}
        del g_big, big, q_big, sums_big

    if distributed:
        # (1) the same loop without the collectives; (2) with one exchange per step; (3) c2 / c3: configs[2]'s objective stack on the
        # same shards, with collectives; (4) a check of the exchange itself: all-reduced packed sums == the sum of the ranks' local ones
        ko_elapsed, _ = measure(plan, 0)
        es_elapsed, _ = measure(plan, 1)
        n_coll_every = n_coll[0]
        c3 = None
        if wl.plan3 is not None:
            if wl.plan3 is not plan:
                measure(wl.plan3, R)
            c3_elapsed, c3_ev = (elapsed, ev_ms) if wl.plan3 is plan else measure(wl.plan3, R)
            c3 = {"value": samples_per_step * args.steps / c3_elapsed, "ms_per_step": c3_elapsed * 1e3 / args.steps,
                  "launch_us": c3_ev * 1e3 / args.steps,
                  "workload": "BASELINE configs[2] objective stack (self-collision + SDF-obstacle + workspace box + "
                              f"EE) on the same shards: {B * world} x {H} over {world} GPUs, with the all-reduce"}
        # what ONE exchange costs end to end when nothing hides it: pack kernel + all-reduce, from enqueue to completion
        torch.cuda.synchronize(dev)
        t_x = time.perf_counter()
        for k in range(10):
            reduce_slot(plan, (slot[0] + k) % n_slots)
            if mailbox is not None:
                mailbox_flush()
            torch.cuda.synchronize(dev)
        exchange_us = (time.perf_counter() - t_x) / 10 * 1e6
        wl.step(bs_ptr, stream.cuda_stream)
        local = torch.zeros(1 + H + H * D, **ta)
        pack_sums(plan, local)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        check_out = local.clone()
        if mailbox is not None:
            mailbox.exchange(local, check_out, stream.cuda_stream)
        elif native is not None:
            native.all_reduce_sum_(check_out, stream.cuda_stream)
        else:
            dist.all_reduce(check_out)
        torch.cuda.synchronize(dev)
        expect = torch.stack(gathered).double().sum(0)
        err = float(((check_out.double() - expect).abs() / (expect.abs() + 1.0)).max().item())
        # the mailbox adds the rows in rank order in fp32: every rank must hold the SAME bits as that sum computed here
        bitwise = None
        if mailbox is not None:
            acc = gathered[0].clone()
            for g_ in gathered[1:]:
                acc += g_
            bitwise = bool(torch.equal(acc, check_out))
        # the all-reduce path measured next to the mailbox (the mailbox switched off for one measurement)
        rccl_cmp = None
        if mailbox is not None:
            mb_keep, mailbox = mailbox, None
            measure(plan, R)
            rc_elapsed, _ = measure(plan, R)
            mailbox = mb_keep
            rccl_cmp = {"value": samples_per_step * args.steps / rc_elapsed, "ms_per_step": rc_elapsed * 1e3 / args.steps,
                        "collectives_in_timed_region": n_coll[0],
                        "via": "librccl ncclAllReduce (ctypes)" if native is not None else "torch.distributed.all_reduce"}
        mb_status = mailbox.status() if mailbox is not None else None
        out["multi_gpu"] = {
            "backend": dist.get_backend(), "ranks": dist.get_world_size(), "reduce_every": args.reduce_every, "reduce_every_effective": R,
            "exchange_via": ("peer-to-peer mailbox (trk_mailbox_exchange: stores into every peer's device memory, rows added in rank order"
                             + ("; captured into the step graphs)" if S_live[0] else "; eager, on the launch stream)")) if mailbox is not None else
                            "librccl ncclAllReduce (ctypes)" if native is not None else "torch.distributed.all_reduce",
            **({"mailbox": {"exchanges": mb_status[0], "timeouts": mb_status[1], "memory": mb_status[2], "sums_bit_identical_to_rank_order": bitwise}}
               if mailbox is not None else {"mailbox": None, "mailbox_note": mailbox_note}),
            **({"rccl_allreduce": rccl_cmp} if rccl_cmp else {}),
            "collectives_in_timed_region": n_coll_value, "exchange_us": exchange_us,
            "allreduce_floats": int(local.numel()),
            "with_allreduce": {"value": value, "ms_per_step": elapsed * 1e3 / args.steps, "per_rank": per_rank_value},
            "kernel_only": {"value": samples_per_step * args.steps / ko_elapsed, "ms_per_step": ko_elapsed * 1e3 / args.steps},
            "every_step": {"value": samples_per_step * args.steps / es_elapsed, "ms_per_step": es_elapsed * 1e3 / args.steps,
                           "collectives_in_timed_region": n_coll_every},
            **({"full_stack_c3": c3} if c3 else {}),
            "allreduce_check": {"ok": bool(err < 1e-5), "max_rel_err": err,
                                "sum_cost_all_ranks": float(check_out[0].item()), "sum_cost_rank0": float(gathered[0][0].item())},
        }
        mg = out["multi_gpu"]
        # north_star: "HBM GB/s vs peak at 1/2/4/8 GPUs" -- every rank's own figure: the step's algorithmic bytes x the timed steps over
        # THAT rank's launch-stream time (HIP events around its timed region, exchange included)
        if per_rank_value:
            step_bytes = wl.bps_step * B * H
            mg["per_rank"] = [{"rank": r, "wall_us": per_rank_value["wall_us"][r], "launch_stream_us": t_us,
                               "achieved_GBps": step_bytes * args.steps / (t_us * 1e-6) / 1e9,
                               "frac": step_bytes * args.steps / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
                              for r, t_us in enumerate(per_rank_value["launch_stream_us"])]
            mg["per_rank_bytes_per_step"] = step_bytes
        # what the exchange leaves of ideal weak scaling: N x (a step without any exchange) / (a step of the timed region with its exchange)
        mg["scaling_bound"] = world * mg["kernel_only"]["ms_per_step"] / mg["with_allreduce"]["ms_per_step"]
        mg["exchange_overhead_us_per_step"] = 1e3 * (mg["with_allreduce"]["ms_per_step"] - mg["kernel_only"]["ms_per_step"])
        assert err < 1e-5, f"all-reduced sums differ from the sum of the ranks' sums: {err}"
        if mb_status is not None and mb_status[1] != 0:
            print(f"[bench] WARNING: {mb_status[1]} mailbox exchanges timed out after the headline measurement (secondary figures may be polluted)", file=sys.stderr)
        assert bitwise is not False, "the mailbox sums differ from the rows added in rank order"

    if rank == 0 and not distributed and args.independent_streams > 1 and wl.random_q is not None:
        # Secondary figure (never `value`): INDEPENDENT batches alternated over three HIP streams.  With one stream a launch waits
        # for the previous one's ~2 us write tail; with three, the next launch's dispatch, kernarg / q fetch and FK overlap it.
        # A planner's iterations depend on each other, so the headline keeps one stream; a server evaluating unrelated batches
        # gets this rate.  Kernels overlap here, so this is a throughput, not a kernel duration.
        ns = args.independent_streams
        streams3 = [torch.cuda.Stream(dev) for _ in range(ns)]
        plans3 = [plan] + [wl.make_plan(wl.random_q(B * H).reshape(B, H, D).contiguous()) for _ in range(ns - 1)]
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

    if rank == 0 and not distributed and args.cpu_seconds > 0:       # CPU baseline at N = 1: here; at N > 1: below, once the ranks are done
        out["cpu_baseline"] = cpu_baseline(wl, args, torch, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None

    if distributed:
        torch.cuda.synchronize(dev)
        if mailbox is not None:
            dist.barrier()                # no rank unmaps a mailbox a peer may still store into
            graphs.clear()
            mailbox.close()
        if native is not None:
            native.close()
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0 and args.cpu_seconds > 0:
            # N > 1: rank 0's host threads are free only now -- the other ranks have left their last barrier and are exiting (they do
            # no CPU arithmetic).  A shorter sample than at N = 1: the line must not wait long for a baseline that N = 1 already carries.
            from oracle import oracle as orc_
            orc_.set_threads(physical_cores())                     # the launcher pinned OMP_NUM_THREADS for the ranks; the baseline uses the host's cores
            out["cpu_baseline"] = cpu_baseline(wl, args, torch, min(args.cpu_seconds, 6.0))
            out["cpu_baseline"]["note"] = f"rank 0 of {world}, after the process group was destroyed; sample bounded to {min(args.cpu_seconds, 6.0):g} s per leg"
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
