"""The peer-to-peer mailbox exchange (csrc/trk_exchange.hip, SURVEY.md 8e's alternative to the all-reduce) and the packed sums it carries.

Two PROCESSES share cuda:0 and map each other's mailbox through hipIpc handles -- the same code path as two GPUs of one node, minus
the xGMI hop.  Checked: the sums equal gloo's all-reduce of the same rows bit for bit (two ranks: a + b in either order), more
exchanges than slots (slot reuse), a hipGraph replay of the exchange, bad arguments.
"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["TRK_ROOT"])
from torch_robotics_amd.distributed import MailboxAllReduce
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
n = 1 + 64 + 64 * 7
mb = MailboxAllReduce(dev, n, n_slots=3)
assert mb.validate()
gen = torch.Generator(device=dev).manual_seed(100 + rank)
out = torch.empty(n, device=dev)
for it in range(11):                                   # > n_slots: every slot is reused several times
    row = torch.randn(n, device=dev, generator=gen) * (10.0 ** (it % 5))
    mb.exchange(row, out)
    ref = row.cpu().clone()
    dist.all_reduce(ref)                               # gloo, on the host: a + b
    assert torch.equal(out.cpu(), ref), (it, float((out.cpu() - ref).abs().max()))
# captured: the sequence number lives in device memory, a replayed exchange is a new exchange
row = torch.randn(n, device=dev, generator=gen)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    mb.exchange(row, out, side.cuda_stream)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
    mb.exchange(row, out, torch.cuda.current_stream(dev).cuda_stream)
ref = row.cpu().clone(); dist.all_reduce(ref)
for it in range(7):
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
# send now, receive later (what a planner does: the wait is then off the critical path); nothing but the order is required
for it in range(5):
    row = torch.randn(n, device=dev, generator=gen)
    mb.send(row)
    filler = torch.randn(1 << 20, device=dev).sum()    # unrelated work between the halves
    mb.recv(out)
    ref = row.cpu().clone(); dist.all_reduce(ref)
    assert torch.equal(out.cpu(), ref)
n_ex, n_to, kind = mb.status()
assert n_ex == 6 + 11 + 1 + 7 + 5 and n_to == 0, (n_ex, n_to)
dist.barrier()
mb.close()
sys.stdout.write(f"rank-{rank}-ok-{kind}\n"); sys.stdout.flush()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_mailbox_two_processes_one_gpu(tmp_path):
    env = dict(os.environ, TRK_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    script = tmp_path / "mailbox_worker.py"
    script.write_text(WORKER)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", str(script)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "rank-0-ok" in p.stdout and "rank-1-ok" in p.stdout, p.stdout[-500:]


@pytest.mark.gpu
def test_mailbox_single_rank_and_argument_checks():
    import ctypes as C
    import torch
    from torch_robotics_amd import _abi
    from torch_robotics_amd._lib import lib
    from torch_robotics_amd.distributed import MailboxAllReduce
    dev = torch.device("cuda:0")
    mb = MailboxAllReduce(dev, 513, rank=0, world=1)
    assert mb.validate(rounds=9)
    x = torch.randn(513, device=dev)
    out = torch.empty_like(x)
    mb.exchange(x, out)
    torch.cuda.synchronize()
    assert torch.equal(out, x)                         # one rank: the sum of one row
    with pytest.raises(ValueError):
        mb.exchange(x[:100], out)
    with pytest.raises(ValueError):
        mb.exchange(x.double(), out)
    assert mb.status()[:2] == (10, 0) and mb.healthy()
    mb.close()
    # the look-ahead rule (include/trk.h): send k + a may precede recv k only for a <= (n_slots - 2) / 2 -- refused, not corrupted
    for slots, ahead in ((2, 0), (3, 0), (4, 1), (6, 2)):
        mb = MailboxAllReduce(dev, 64, n_slots=slots, rank=0, world=1)
        y = torch.arange(64, device=dev, dtype=torch.float32)
        with pytest.raises(ValueError, match="nothing has been sent"):
            mb.recv(out[:64].contiguous())
        for a in range(ahead + 1):
            mb.send(y + a)
        with pytest.raises(ValueError, match="ahead of their receives"):
            mb.send(y)
        o64 = torch.empty(64, device=dev)
        for a in range(ahead + 1):                     # the receives come back in order
            mb.recv(o64)
            torch.cuda.synchronize()
            assert torch.equal(o64, y + a)
        mb.send(y)                                     # ... and the window is open again
        mb.recv(o64)
        assert mb.healthy()
        mb.close()
    L, h = lib(), C.c_void_p()
    for world, rank, n, slots in ((0, 0, 8, 4), (17, 0, 8, 4), (2, 2, 8, 4), (2, 0, 0, 4), (2, 0, 8, 1), (2, 0, 8, 65), (2, 0, (1 << 22) + 1, 4)):
        assert L.trk_mailbox_create(world, rank, n, slots, C.byref(h)) == _abi.TRK_ERR_INVALID_ARG
    assert L.trk_mailbox_exchange(None, None, None, None) == _abi.TRK_ERR_INVALID_ARG
    # a two-rank mailbox that was never connected refuses to exchange
    assert L.trk_mailbox_create(2, 0, 8, 4, C.byref(h)) == 0
    assert L.trk_mailbox_exchange(h, x.data_ptr(), out.data_ptr(), None) == _abi.TRK_ERR_INVALID_ARG
    assert b"connect" in L.trk_last_error()
    L.trk_mailbox_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,floats", [(2, 513), (3, 1921)])
def test_mailbox_soak_small(ranks, floats):
    """tools/mailbox_soak.py with a few thousand exchanges: every rank computes every expected sum by itself (rank-order fp32 sum of
    rows it can regenerate), eager send / receive pairs and replayed graphs with the receive one exchange behind its send; no
    mismatch, no time-out.  (The long form -- 50 000 exchanges on 2 ranks, 12 500 on 4 -- ran clean in round 5.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29580 + ranks), str(ROOT / "tools" / "mailbox_soak.py"), "--single-device", "--exchanges", "3000",
                        "--floats", str(floats)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "mismatching exchanges 0, time-outs 0" in p.stdout


SHARDED_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["TRK_ROOT"])
import torch_robotics_amd as tra
from torch_robotics_amd.distributed import ShardedRollout, shard_batch
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
TA = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=TA)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
B, H = 96, 64
q_all = robot.random_q(B * H, generator=torch.Generator(device=dev).manual_seed(5)).reshape(B, H, 7)      # the same on every rank
lo, hi = shard_batch(B, rank, world)
plan = task.rollout_plan(q_all[lo:hi].contiguous(), w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False)
ref_plan = task.rollout_plan(q_all.contiguous(), w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False)
ref_plan.launch(); torch.cuda.synchronize()
ref = torch.cat([ref_plan.cost.double().sum().reshape(1), ref_plan.cost.double().sum(0), ref_plan.gq.double().sum(0).reshape(-1)])
for mode in ("auto", "allreduce"):
    sh = ShardedRollout(plan, exchange=mode)
    assert (sh.mailbox is not None) == (mode == "auto"), sh.mailbox_note
    for it in range(3):
        sh.launch()
        tot = sh.exchange().clone()
        torch.cuda.synchronize()
        assert float((tot.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (mode, it)
        # sharding invariance of the per-sample outputs: this rank's block equals the same rows of the unsharded evaluation
        assert torch.equal(plan.cost, ref_plan.cost[lo:hi]) and torch.equal(plan.gq, ref_plan.gq[lo:hi])
    # every rank holds the same bits
    mine = tot.cpu(); other = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    assert all(torch.equal(o, other[0]) for o in other)
    sh.launch(); sh.send()
    try:
        sh.send(); raise SystemExit("expected a RuntimeError")
    except RuntimeError:
        pass
    sh.recv(); torch.cuda.synchronize()
    dist.barrier(); sh.close()
sys.stdout.write(f"sharded-{rank}-ok\n"); sys.stdout.flush()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_sharded_rollout_two_processes_one_gpu(tmp_path):
    """distributed.ShardedRollout: two ranks evaluate their blocks of one batch and exchange the packed sums (mailbox, and the
    all-reduce path): the totals equal the unsharded evaluation's sums, every rank holds the same bits, the per-sample outputs of a
    block equal the same rows of the unsharded evaluation bit for bit."""
    env = dict(os.environ, TRK_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    script = tmp_path / "sharded_worker.py"
    script.write_text(SHARDED_WORKER)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29591", str(script)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "sharded-0-ok" in p.stdout and "sharded-1-ok" in p.stdout


TIMEOUT_WORKER = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["TRK_ROOT"])
from torch_robotics_amd.distributed import MailboxAllReduce
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
n = 257
# the ranks must agree on the mailbox's shape: a mismatch is refused on EVERY rank before any peer memory is mapped
try:
    MailboxAllReduce(dev, n + rank, n_slots=4)
    raise SystemExit("expected a RuntimeError: the ranks disagree on n_floats")
except RuntimeError as e:
    assert "disagree" in str(e)
mb = MailboxAllReduce(dev, n, n_slots=4)
assert mb.validate()
dist.barrier()
row = torch.ones(n, device=dev)
out = torch.zeros(n, device=dev)
if rank == 0:                       # rank 1 never sends this exchange: a dead / late peer
    mb.send(row); mb.recv(out)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all()), out[:4]                 # NaN everywhere: never a plausible sum with stale rows in it
    assert not mb.healthy() and mb.status()[1] == 1
dist.barrier()
mb.close()
sys.stdout.write(f"timeout-{rank}-ok\n"); sys.stdout.flush()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_mailbox_timeout_writes_nan_and_shapes_must_agree(tmp_path):
    """a receive that times out must not hand back a sum that includes a late peer's stale rows; the ranks' (world, n_floats,
    n_slots) are compared before any peer memory is mapped."""
    env = dict(os.environ, TRK_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", TRK_MAILBOX_TIMEOUT_S="0.5")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    script = tmp_path / "mailbox_timeout_worker.py"
    script.write_text(TIMEOUT_WORKER)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29567", str(script)], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "timeout-0-ok" in p.stdout and "timeout-1-ok" in p.stdout, p.stdout[-500:]
