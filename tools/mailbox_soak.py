#!/usr/bin/env python3
"""Soak test of the peer-to-peer mailbox: N processes (one per GPU, or all on cuda:0 with --single-device) exchange many rows whose
sums every rank can compute by itself -- row_r(k) = cos((i + 1) * (k % 977 + 1) * (r + 1) * 1e-3) -- eagerly and as replayed graphs of
16 exchanges, and count mismatches and time-outs.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 tools/mailbox_soak.py --single-device [--exchanges 50000]
"""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from torch_robotics_amd.distributed import MailboxAllReduce

ap = argparse.ArgumentParser()
ap.add_argument("--exchanges", type=int, default=50000)
ap.add_argument("--floats", type=int, default=513)
ap.add_argument("--single-device", action="store_true")
ap.add_argument("--backend", default="gloo")
a = ap.parse_args()
dist.init_process_group(a.backend)
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0 if a.single_device else int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
mb = MailboxAllReduce(dev, a.floats, n_slots=4)
assert mb.validate()
idx = torch.arange(1, a.floats + 1, device=dev, dtype=torch.float32)


def row(r, k):
    return torch.cos(idx * float((k % 977 + 1) * (r + 1)) * 1e-3)


def expect(k):
    acc = row(0, k)
    for r in range(1, world):
        acc = acc + row(r, k)
    return acc


bad = torch.zeros((), device=dev, dtype=torch.int64)
out = torch.empty(a.floats, device=dev)
t0 = time.perf_counter()
for k in range(a.exchanges):
    mb.send(row(rank, k))
    mb.recv(out)
    bad += (out != expect(k)).any()
    if k % 5000 == 4999:
        torch.cuda.synchronize()
torch.cuda.synchronize()
t_eager = time.perf_counter() - t0
# graphs of 16 exchanges: rows are graph inputs, refilled before each replay
G = 16
rows_in = torch.empty(G, a.floats, device=dev)
outs = torch.empty(G, a.floats, device=dev)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    for j in range(G):
        rows_in[j] = row(rank, j)
        mb.send(rows_in[j], side.cuda_stream); mb.recv(outs[j], side.cuda_stream)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
    cur = torch.cuda.current_stream(dev).cuda_stream
    for j in range(G):
        mb.send(rows_in[j], cur)
        if j > 0:
            mb.recv(outs[j - 1], cur)           # a receive one exchange behind its send, like the bench's step graphs: ONE send of
                                                # look-ahead, which needs n_slots >= 4 (include/trk.h: a <= (n_slots - 2) / 2); the library refuses more
    mb.recv(outs[G - 1], cur)
n_rep = max(1, a.exchanges // (4 * G))
t0 = time.perf_counter()
for rep in range(n_rep):
    base = 1000 + rep * G
    for j in range(G):
        rows_in[j] = row(rank, base + j)
    g.replay()
    for j in range(G):
        bad += (outs[j] != expect(base + j)).any()
torch.cuda.synchronize()
t_graph = time.perf_counter() - t0
n_ex, n_to, kind = mb.status()
res = torch.tensor([int(bad.item()), n_to], dtype=torch.int64)
dist.all_reduce(res)
dist.barrier()
mb.close()
if rank == 0:
    print(f"mailbox soak: {world} ranks, {a.floats} floats, {a.exchanges} eager exchanges in {t_eager:.1f} s, {n_rep} graph replays x {G} in "
          f"{t_graph:.1f} s, memory {kind}: mismatching exchanges {int(res[0])}, time-outs {int(res[1])}", flush=True)
dist.destroy_process_group()
sys.exit(0 if int(res[0]) == 0 and int(res[1]) == 0 else 1)
