"""bench.py's N > 1 control flow.

CPU: `bench.py --gpus 2` without a launcher must start the two ranks itself as a child `torch.distributed.run` (before it touches
the GPU) and hand the child's failure through as its own exit code -- in the GPU-less build container the ranks die at
`torch.cuda.set_device`, which is exactly the failure path.
GPU: the same command in its debug mode (`--dist-backend gloo --single-device`: two ranks share cuda:0, the collectives go through
gloo) runs the whole sharded loop on a 1-GPU box; the JSON line is checked, incl. the all-reduce of the packed sums.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(extra, timeout):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + extra, env=env, cwd=str(ROOT), capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_spawns_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("failure path of the launcher: needs a box without a GPU")
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"], 300)
    assert p.returncode != 0
    # both ranks were started by the elastic launcher and reported as failed children
    assert "torch.distributed" in p.stderr and "ChildFailedError" in p.stderr, p.stderr[-2000:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_launcher_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, cwd=str(ROOT), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=3" in p.stderr


@pytest.mark.gpu
def test_two_rank_bench_on_one_gpu_gloo():
    p = _run(["--gpus", "2", "--dist-backend", "gloo", "--single-device", "--same-q", "--steps", "20", "--warmup", "5",
              "--reduce-every", "8", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 20
    assert out["config"]["reduce_every"] == 8 and out["config"]["global_batch"] == 8192
    mg = out["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["ranks"] == 2 and mg["collectives_in_timed_region"] in (2, 3) and mg["exchange_us"] > 0
    assert mg["allreduce_floats"] == 1 + 64 + 64 * 7
    chk = mg["allreduce_check"]
    assert chk["ok"] and chk["max_rel_err"] < 1e-5
    # --same-q: both ranks evaluate the same trajectories, so the all-reduced cost sum is exactly twice rank 0's
    assert chk["sum_cost_all_ranks"] == pytest.approx(2.0 * chk["sum_cost_rank0"], rel=1e-6)
    for k in ("with_allreduce", "kernel_only", "every_step", "full_stack_c3"):
        assert mg[k]["value"] > 0 and mg[k]["ms_per_step"] > 0
    assert mg["every_step"]["collectives_in_timed_region"] == 20 and mg["reduce_every_effective"] == 8
    assert out["value"] == mg["with_allreduce"]["value"]
    assert out["cpu_baseline"] is None and out["roofline"]["bytes_per_sample"] == 192


@pytest.mark.gpu
def test_two_rank_config5_on_one_gpu_gloo():
    """configs[4] as a driver-runnable workload, N-rank: dual Panda, fp16 I/O with the loss scale, GP prior; the exchange packs the
    fp16 gradient (unscaled, fp32) and the GP cost.  Two ranks share cuda:0 over gloo; the driver's 5 + 20 steps at the DEFAULT cadence
    still contain exchanges."""
    p = _run(["--config", "c5", "--gpus", "2", "--dist-backend", "gloo", "--single-device", "--same-q", "--steps", "20", "--warmup", "5",
              "--batch", "256", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["dtype"] == "f16" and out["config"]["objectives"] == "c5" and out["config"]["horizon"] == 128
    assert out["config"]["global_batch"] == 512 and out["config"]["launches_per_step"] == 1 and out["config"]["kernel"] == "specialized"
    assert 0 < out["config"]["grad_scale"] < 1
    # ONE launch: q, qd in (28 + 28), positions (138), cost (4), gq, gqd out (28 + 28)
    assert out["roofline"]["bytes_per_sample"] == 254 and "step" not in out["roofline"]
    mg = out["multi_gpu"]
    assert mg["reduce_every"] == 64 and mg["reduce_every_effective"] == 20 and mg["collectives_in_timed_region"] == 1
    assert mg["allreduce_floats"] == 1 + 128 + 128 * 14 and "full_stack_c3" not in mg
    chk = mg["allreduce_check"]
    assert chk["ok"] and chk["sum_cost_all_ranks"] == pytest.approx(2.0 * chk["sum_cost_rank0"], rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,dtype,launches,extra", [("c4", "f32", 1, []), ("c4", "f32", 2, ["--two-launch"]), ("c5", "f16", 1, []),
                                                      ("c5", "f16", 2, ["--two-launch"])])
def test_bench_configs_4_and_5_one_gpu(cfg, dtype, launches, extra):
    """`bench.py --config c4|c5` on one GPU: a driver-parsable line with the config's own bytes, the dominant kernel's roofline,
    the step's, and a CPU baseline from the oracle (incl. the Jacobian / the GP term)."""
    p = _run(["--config", cfg, "--steps", "20", "--warmup", "5", "--cpu-seconds", "1.5", "--batch", "512"] + extra, 900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dtype"] == dtype and out["config"]["launches_per_step"] == launches and out["config"]["kernel"] == "specialized"
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["unit"] == "rollouts/s" and out["vs_baseline"] is None
    D, L, e = (22, 30, 4) if cfg == "c4" else (14, 23, 2)
    # one-launch forms move the step's extra bytes in the dominant kernel: c5 qd in + gqd out, c4 pos / quat / lin_jac / ang_jac out
    fused_extra = (4 * D if cfg == "c5" else 28 + 24 * D) if launches == 1 else 0
    assert out["roofline"]["bytes_per_sample"] == 2 * e * D + 3 * e * L + 4 + fused_extra
    assert ("Jacobian" in out["roofline"]["kernel"]) == (cfg == "c4" and launches == 1)
    assert 0 < out["roofline"]["frac"] < 1
    if launches > 1:
        assert 0 < out["roofline"]["step"]["frac"] < 1 and out["roofline"]["step"]["step_us"] >= out["roofline"]["launch_us"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["one_core"]["value"] > 0
    assert ("Jacobian" in cb["sample"]) == (cfg == "c4") and ("GP prior" in cb["sample"]) == (cfg == "c5")


@pytest.mark.gpu
@pytest.mark.parametrize("scene,bps", [("grid", 272), ("shelf", 192), ("maze", 192)])
def test_bench_scene_variants(scene, bps):
    p = _run(["--scene", scene, "--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--batch", "512"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["scene"] == scene and out["roofline"]["bytes_per_sample"] == bps
    assert out["config"]["kernel"] == "specialized" and out["value"] > 0


@pytest.mark.gpu
def test_rccl_code_path_on_one_rank():
    """`--force-dist`: the N > 1 code path with ONE rank and the real backend ("nccl" = RCCL on ROCm): process-group creation on
    the device, the side-stream all-reduce of the packed sums, the in-stream barriers, the MAX over ranks, all_gather -- on hardware."""
    p = _run(["--force-dist", "--exchange", "rccl", "--steps", "40", "--warmup", "5", "--reduce-every", "8", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 1 and mg["backend"] == "nccl" and mg["ranks"] == 1 and mg["exchange_via"].startswith("torch.distributed")
    assert mg["collectives_in_timed_region"] == 5 and mg["allreduce_check"]["ok"] and mg["exchange_us"] > 0
    assert mg["allreduce_check"]["sum_cost_all_ranks"] == pytest.approx(mg["allreduce_check"]["sum_cost_rank0"], rel=1e-6)
    # the driver's settings (20 steps after 5) at the default cadence (64): the interval shrinks to the run, the region contains
    # an exchange -- `value` is never a kernel-only figure
    p = _run(["--force-dist", "--exchange", "rccl", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["multi_gpu"]["collectives_in_timed_region"] == 1 and out["config"]["reduce_every"] == 64
    assert out["config"]["reduce_every_effective"] == 20 and out["multi_gpu"]["every_step"]["collectives_in_timed_region"] == 20


@pytest.mark.gpu
def test_native_rccl_exchange_on_one_rank():
    """`--native-rccl`: the exchange issued straight on librccl (ncclAllReduce through ctypes on the side stream)."""
    p = _run(["--force-dist", "--exchange", "rccl", "--native-rccl", "--steps", "40", "--warmup", "5", "--reduce-every", "8", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert mg["exchange_via"].startswith("librccl") and mg["collectives_in_timed_region"] == 5 and mg["allreduce_check"]["ok"]
    assert mg["allreduce_check"]["sum_cost_all_ranks"] == pytest.approx(mg["allreduce_check"]["sum_cost_rank0"], rel=1e-6)


@pytest.mark.gpu
def test_launch_modes_graph_by_default_eager_on_request():
    """The step loop is captured by default (hipGraph replays of <= 100 evaluations; W and K stay exact: 7 + 23 steps need graphs of
    7 and 23), `--graph 0` is the eager loop, c4's two-kernel step and the gloo debug mode default to eager; both modes leave the same
    outputs behind (bench.py's own closing check compares the cost sums with the per-sample costs)."""
    seen = {}
    for mode, extra in (("graph", []), ("eager", ["--graph", "0"]), ("small graphs", ["--graph", "4"])):
        p = _run(["--steps", "23", "--warmup", "7", "--cpu-seconds", "0", "--batch", "512", "--no-out-of-cache"] + extra, 600)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert out["steps"] == 23 and out["warmup"] == 7 and out["value"] > 0 and "capture failed" not in p.stderr
        seen[mode] = out["config"]["launch"]
    assert seen["graph"].startswith("hipGraph replays of <= 100") and seen["eager"].startswith("eager")
    assert seen["small graphs"].startswith("hipGraph replays of <= 4")
    p = _run(["--config", "c4", "--two-launch", "--steps", "10", "--warmup", "2", "--cpu-seconds", "0", "--batch", "256"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["config"]["launch"].startswith("eager")
    p = _run(["--config", "c4", "--steps", "10", "--warmup", "2", "--cpu-seconds", "0", "--batch", "256"], 600)      # one launch per step: captured
    assert p.returncode == 0, p.stderr[-3000:]
    assert json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["config"]["launch"].startswith("hipGraph")
    # N > 1 code path (one rank, RCCL) in graph mode: the exchange sits between two replays inside the timed region
    p = _run(["--force-dist", "--exchange", "rccl", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--batch", "512"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["launch"].startswith("hipGraph") and out["multi_gpu"]["collectives_in_timed_region"] == 1
    assert out["multi_gpu"]["allreduce_check"]["ok"]


@pytest.mark.gpu
def test_mailbox_exchange_is_the_default_and_rides_in_the_graph():
    """`--exchange auto` (default): the peer-to-peer mailbox passes its known-answer validation and carries the exchange; at the
    driver's settings the 20 timed steps AND their exchange are one captured graph; the all-reduce path is measured next to it."""
    p = _run(["--force-dist", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert mg["exchange_via"].startswith("peer-to-peer mailbox") and "captured" in mg["exchange_via"]
    assert mg["collectives_in_timed_region"] == 1 and mg["every_step"]["collectives_in_timed_region"] == 20
    assert mg["mailbox"]["timeouts"] == 0 and mg["mailbox"]["exchanges"] > 40 and mg["mailbox"]["sums_bit_identical_to_rank_order"]
    assert mg["allreduce_check"]["ok"] and mg["rccl_allreduce"]["ms_per_step"] > 0 and mg["rccl_allreduce"]["collectives_in_timed_region"] == 1
    assert 0 < mg["scaling_bound"] <= 1.05 and mg["exchange_overhead_us_per_step"] < 3.0
    # eager loop: the same exchange on a side stream
    p = _run(["--force-dist", "--exchange", "p2p", "--graph", "0", "--steps", "24", "--warmup", "4", "--reduce-every", "8", "--cpu-seconds", "0",
              "--batch", "512"], 600)
    assert p.returncode == 0, p.stderr[-3000:]
    mg = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["multi_gpu"]
    assert "eager" in mg["exchange_via"] and mg["collectives_in_timed_region"] == 3 and mg["mailbox"]["timeouts"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,extra", [("c2", []), ("c2", ["--graph", "20"]), ("c5", ["--batch", "256", "--graph", "20"])])
def test_two_rank_mailbox_exchange_on_one_gpu(cfg, extra):
    """Two processes share cuda:0 and exchange through each other's mailboxes (hipIpc handles): the sums every rank reads equal the
    rows added in rank order bit for bit, equal gloo's all-reduce of the same rows, nothing times out -- eager and captured."""
    p = _run(["--config", cfg, "--gpus", "2", "--dist-backend", "gloo", "--single-device", "--exchange", "p2p", "--steps", "20", "--warmup", "5",
              "--reduce-every", "8", "--cpu-seconds", "0"] + extra, 600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert mg["ranks"] == 2 and mg["exchange_via"].startswith("peer-to-peer mailbox") and ("captured" in mg["exchange_via"]) == bool(extra and "--graph" in extra)
    assert mg["mailbox"]["timeouts"] == 0 and mg["mailbox"]["sums_bit_identical_to_rank_order"] and mg["allreduce_check"]["ok"]
    assert mg["collectives_in_timed_region"] in (2, 3) and mg["every_step"]["collectives_in_timed_region"] == 20


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,ranks,extra", [("c2", 8, ["--batch", "256"]), ("c5", 4, ["--batch", "128"])])
def test_many_rank_dress_rehearsal_on_one_gpu(cfg, ranks, extra):
    """The 8-GPU run (and config 5's 4-GPU split) rehearsed on ONE GPU: N processes share cuda:0, the packed sums travel through the
    world-N mailbox (hipIpc handles, the N-rank argument struct, slot reuse), captured into the step graphs; the line carries what a
    SCALE line must carry -- `roofline`, `cpu_baseline` (rank 0, after the process group is gone), per-rank bandwidth fractions."""
    p = _run(["--config", cfg, "--gpus", str(ranks), "--single-device", "--dist-backend", "gloo", "--exchange", "p2p", "--graph", "20",
              "--steps", "20", "--warmup", "5", "--cpu-seconds", "1.5"] + extra, 1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert out["n_gpus"] == ranks and mg["ranks"] == ranks and out["scaling"] == "weak"
    assert mg["exchange_via"].startswith("peer-to-peer mailbox") and "captured" in mg["exchange_via"]
    assert mg["mailbox"]["timeouts"] == 0 and mg["mailbox"]["sums_bit_identical_to_rank_order"] and mg["allreduce_check"]["ok"]
    assert mg["collectives_in_timed_region"] == 1 and mg["every_step"]["collectives_in_timed_region"] == 20
    pr = mg["per_rank"]
    assert [r["rank"] for r in pr] == list(range(ranks))
    for r in pr:
        assert r["launch_stream_us"] > 0 and r["achieved_GBps"] > 0 and 0 < r["frac"] < 1
        assert r["achieved_GBps"] == pytest.approx(mg["per_rank_bytes_per_step"] * 20 / (r["launch_stream_us"] * 1e-6) / 1e9, rel=1e-2)
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["bytes_per_sample"] == (192 if cfg == "c2" else 254)
    cb = out["cpu_baseline"]
    assert cb is not None and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "rank 0" in cb["note"]


def test_exchange_schedule_is_exact():
    """bench.exchange_schedule (pure): whatever the step count, cadence, graph size and mode, a run issues exactly `count` steps and
    exactly one exchange per `cadence` steps (the ones of exchange_pieces), graphs hold at most `graph_steps` steps and `n_slots`
    exchanges, and the driver's 5 + 20 steps in mailbox mode are one plain graph and ONE graph with the exchange in its middle."""
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    assert bench.exchange_schedule(20, 20, 100, True, 128) == [("graph", (10, 10))]
    assert bench.exchange_schedule(5, 20, 100, True, 128) == [("graph", (5,))]
    assert bench.exchange_schedule(20, 20, 100, False, 128) == [("graph", (10,)), ("exchange",), ("graph", (10,))]
    assert bench.exchange_schedule(20, 20, 0, True, 128) == [("steps", 10), ("exchange",), ("steps", 10)]
    assert bench.exchange_schedule(20, 1, 100, True, 128) == [("graph", (1,) * 20 + (0,))]
    assert bench.exchange_schedule(0, 8, 100, True, 128) == []
    for count in (1, 5, 20, 23, 64, 100, 101, 777, 2000):
        for cadence in (0, 1, 2, 7, 8, 20, 64, 500):
            want_ex = sum(1 for _, ex in bench.exchange_pieces(count, cadence) if ex)
            assert sum(n for n, _ in bench.exchange_pieces(count, cadence)) == count
            assert want_ex == (0 if not cadence else len([j for j in range(1, count + 1) if j % cadence == cadence // 2]))
            for S in (0, 4, 20, 100):
                for mailbox in (False, True):
                    for n_slots in (4, 128):
                        items = bench.exchange_schedule(count, cadence, S, mailbox, n_slots)
                        steps = sum(sum(it[1]) if it[0] == "graph" else (it[1] if it[0] == "steps" else 0) for it in items)
                        exch = sum((len(it[1]) - 1) if it[0] == "graph" else (1 if it[0] == "exchange" else 0) for it in items)
                        assert steps == count and exch == want_ex, (count, cadence, S, mailbox, n_slots, items)
                        for it in items:
                            if it[0] == "graph":
                                assert sum(it[1]) <= S and len(it[1]) - 1 <= n_slots and (sum(it[1]) > 0 or len(it[1]) > 1)
                                assert mailbox or len(it[1]) == 1          # all-reduce mode: graphs hold plain steps only
                            assert S or it[0] != "graph"
