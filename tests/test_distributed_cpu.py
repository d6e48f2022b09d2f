"""N > 1 path on CPU: 2 processes, gloo.  Each rank evaluates its contiguous block of whole trajectories
(with the CPU oracle standing in for the GPU kernel, as allowed for tests), packs its partial sums and
all-reduces them; the result must equal the unsharded evaluation.  Also checks the sharding helper."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

from torch_robotics_amd.distributed import shard_batch

ROOT = Path(__file__).resolve().parent.parent


def test_shard_batch_partitions_exactly():
    for n in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            blocks = [shard_batch(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_batch(8, 2, 2)


WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["TRK_ROOT"]); sys.path.insert(0, os.path.join(os.environ["TRK_ROOT"], "tests"))
    from helpers import gold, model, panda_cost_spec
    from oracle.oracle import Oracle
    from torch_robotics_amd.distributed import shard_batch, all_reduce_sum_
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['TRK_PORT']}",
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    g, robot, gs = gold("rollout_panda"), gold("panda_robot"), gold("cost_spheres3d")
    o = Oracle(model("panda_arm_no_gripper"), panda_cost_spec(gs, robot, ee_target=g["target"]))
    q = g["q"]                                    # (6, 64, 7): six trajectories
    lo, hi = shard_batch(q.shape[0], rank, world)
    _, cost, gq = o.rollout(q[lo:hi].reshape(-1, 7), (1, 1, 1, 1), "f32")
    cost = cost.reshape(hi - lo, 64); gq = gq.reshape(hi - lo, 64, 7)
    # packed partials: [sum cost | sum_b cost(h) (64) | sum_b grad (64*7)]
    packed = torch.from_numpy(np.concatenate([[cost.sum()], cost.sum(0), gq.sum(0).reshape(-1)]).astype(np.float32))
    all_reduce_sum_(packed)
    if rank == 0:
        _, c_all, g_all = o.rollout(q.reshape(-1, 7), (1, 1, 1, 1), "f32")
        c_all = c_all.reshape(6, 64); g_all = g_all.reshape(6, 64, 7)
        ref = np.concatenate([[c_all.sum()], c_all.sum(0), g_all.sum(0).reshape(-1)])
        err = np.abs(packed.numpy() - ref).max() / np.abs(ref).max()
        assert err < 1e-5, err
        assert np.allclose(cost, c_all[lo:hi])     # per-sample outputs stay sharded and equal the unsharded slice
        print("OK", err)
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_rank_gloo_shard_sum_equals_unsharded(tmp_path, oracle_lib):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", TRK_PORT=str(port), TRK_ROOT=str(ROOT),
                   OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]
