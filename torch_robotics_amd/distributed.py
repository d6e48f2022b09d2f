"""Multi-GPU use of the hot path: one process per GPU, batch sharding, one tiny collective.

Every (batch, horizon) sample is independent for FK and for the collision / EE objectives, and the terms that
couple time steps stay inside one trajectory, so the batch dimension is split in contiguous blocks of whole
trajectories.  Models, cost tables and the SDF grid are replicated.  The only exchange
step is the sum of the per-rank packed partial sums [sum cost | cost per time step | gradient per time step and joint]: either a
PEER-TO-PEER MAILBOX (`MailboxAllReduce`: every rank stores its row into every peer's device memory over xGMI and adds the rows in
rank order; capturable into a hipGraph) or one small all-reduce -- RCCL over xGMI on the GPU box (`backend="nccl"`), gloo in the
tests.  `ShardedRollout` is the whole thing as one object (evaluate this rank's block, send, receive).  Per-sample outputs (cost,
gradient, link positions) stay sharded.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_batch(n_trajectories: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) block of trajectories owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(n_trajectories), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_reduce_sum_(packed: torch.Tensor, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
    """In-place sum of a small packed buffer of partial sums over all ranks (no-op without a process group)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class CostSumReducer:
    """Folds the fused kernel's per-wavefront cost sums into one scalar per evaluation and all-reduces it on a
    side stream, so the launch stream never waits for the collective."""

    def __init__(self, device, n_slots: int = 64):
        from . import ops
        self._ops = ops
        self.device = torch.device(device)
        self.slots = torch.zeros(n_slots, device=self.device, dtype=torch.float32)
        self.side = torch.cuda.Stream(self.device)
        self._k = 0

    def submit(self, block_sums: torch.Tensor) -> torch.Tensor:
        """Returns a 1-element view that holds the global cost sum once the side stream has run."""
        k = self._k % self.slots.numel()
        self._k += 1
        out = self.slots[k:k + 1]
        self._ops.reduce_sum(block_sums, out=out)              # deterministic, on the current stream
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            all_reduce_sum_(out)
        return out

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.side)


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class RcclAllReduce:
    """The planner's one exchange -- an in-place fp32 sum of a small packed buffer over all ranks -- issued straight on librccl
    (`ncclAllReduce` on a caller-chosen HIP stream) instead of through `torch.distributed`: the collective itself is the same RCCL
    ring over xGMI, but enqueueing it costs one ctypes call instead of c10d's ~50 us of host work per call, which matters when the
    exchange is 2 kB and an evaluation is 10 us.  The communicator is this class's own: rank 0 creates a `ncclUniqueId` and the
    already initialised `torch.distributed` group (any backend) broadcasts its 128 bytes.  Opt-in (`bench.py --native-rccl`);
    `all_reduce_sum_` remains the default path.
    Two communicators on one device must not have collectives in flight at the same time (RCCL / NCCL: deadlock-prone): a caller
    that also uses `torch.distributed` on the device orders them -- the stream of one waits for the other's (`bench.py`: the
    in-stream barriers wait for the side stream this class's all-reduces run on)."""

    NCCL_FLOAT32, NCCL_SUM = 7, 0

    def __init__(self, device, rank: Optional[int] = None, world: Optional[int] = None):
        self.device = torch.device(device)
        lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._lib = L = C.CDLL(lib_path)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        have_group = dist.is_available() and dist.is_initialized()
        self.rank = int(rank if rank is not None else (dist.get_rank() if have_group else 0))
        self.world = int(world if world is not None else (dist.get_world_size() if have_group else 1))
        uid = _NcclUniqueId()
        if self.rank == 0:
            self._check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            if not have_group:
                raise RuntimeError("RcclAllReduce: more than one rank needs an initialised torch.distributed group to share the id")
            box = [bytes(uid.internal) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            C.memmove(C.byref(uid), box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what}: {self._lib.ncclGetErrorString(rc).decode()} (ncclResult {rc})")

    def all_reduce_sum_(self, buf: torch.Tensor, stream: Optional[int] = None) -> None:
        """In-place sum of a contiguous fp32 device buffer over the ranks, enqueued on `stream` (a raw HIP stream handle; default:
        torch's current stream on the device).  Asynchronous like a kernel launch."""
        if buf.device != self.device or buf.dtype != torch.float32 or not buf.is_contiguous():
            raise ValueError(f"RcclAllReduce: expected a contiguous float32 tensor on {self.device}")
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        p = buf.data_ptr()
        rc = self._lib.ncclAllReduce(p, p, buf.numel(), self.NCCL_FLOAT32, self.NCCL_SUM, self._comm, stream)
        if rc:
            self._check(rc, "ncclAllReduce")

    def close(self) -> None:
        comm, self._comm = self._comm, None
        if comm:
            self._lib.ncclCommDestroy(comm)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MailboxAllReduce:
    """The planner's one exchange as a PEER-TO-PEER MAILBOX (`trk_mailbox_*`, csrc/trk_exchange.hip; SURVEY.md 8e's alternative to
    the all-reduce): every rank stores its packed row of partial sums straight into a slot of every peer's mailbox -- device memory
    mapped through hipIpc handles, one xGMI hop, all peers in parallel --, raises a sequence flag, waits for the flags of its own
    mailbox and adds the rows in RANK ORDER (bit-identical on every rank and on every run).  Two single-workgroup kernels per
    exchange (`send`: stores + flag, never waits; `recv`: wait + sum -- a planner issues it a few evaluations later, when the wait is
    free), capturable into a hipGraph; no host work, no collective library on the data path.  The handles travel once, at
    construction, through the already initialised `torch.distributed` group (any backend).

        mb = MailboxAllReduce(device, n_floats)          # collective over the group: every rank constructs it
        mb.exchange(packed, out, stream)                  # out = sum over ranks of packed; asynchronous

    `validate()` runs a known-answer exchange on all ranks and returns False (instead of hanging) when the peer stores do not
    arrive -- the caller then keeps the RCCL path (`all_reduce_sum_`)."""

    ALLOC_KINDS = ("uncached", "fine-grained", "plain")

    def __init__(self, device, n_floats: int, n_slots: int = 4, group: Optional[dist.ProcessGroup] = None,
                 rank: Optional[int] = None, world: Optional[int] = None):
        from ._lib import check, lib
        self._lib, self._check = lib(), check
        self.device = torch.device(device)
        have_group = dist.is_available() and dist.is_initialized()
        self.rank = int(rank if rank is not None else (dist.get_rank(group) if have_group else 0))
        self.world = int(world if world is not None else (dist.get_world_size(group) if have_group else 1))
        self.n_floats, self.group = int(n_floats), group
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.trk_mailbox_create(self.world, self.rank, self.n_floats, int(n_slots), C.byref(self._h))
            err = None if rc == 0 else self._lib.trk_last_error().decode("utf-8", "replace")
            handle = C.create_string_buffer(64)
            if err is None and self.world > 1:
                rc = self._lib.trk_mailbox_ipc_handle(self._h, handle)
                err = None if rc == 0 else self._lib.trk_last_error().decode("utf-8", "replace")
            if self.world > 1:
                # every rank takes part in the gather even when its own allocation failed: the others must not wait for it forever
                if not have_group:
                    raise RuntimeError("MailboxAllReduce: more than one rank needs an initialised torch.distributed group to share the handles")
                box = [None] * self.world
                shape = (self.world, self.n_floats, int(n_slots))
                dist.all_gather_object(box, (err, bytes(handle.raw), shape), group=group)
                errs = [e for e, _, _ in box if e is not None]
                if errs:
                    self.close()
                    raise RuntimeError(f"MailboxAllReduce: a rank could not create its mailbox: {errs[0]}")
                # the row and flag offsets inside a PEER's mailbox follow from (world, n_floats, n_slots): ranks that disagree would
                # store past each other's rows -- silent corruption of another process's device memory.  Every rank sees the same
                # list, so every rank raises.
                if any(sh != shape for _, _, sh in box):
                    self.close()
                    raise RuntimeError(f"MailboxAllReduce: the ranks disagree on (world, n_floats, n_slots): {[sh for _, _, sh in box]}")
                blob = b"".join(h for _, h, _ in box)
                rc = self._lib.trk_mailbox_connect(self._h, blob)
                err = None if rc == 0 else self._lib.trk_last_error().decode("utf-8", "replace")
                box2 = [None] * self.world
                dist.all_gather_object(box2, err, group=group)
                errs = [e for e in box2 if e is not None]
                if errs:
                    self.close()
                    raise RuntimeError(f"MailboxAllReduce: a rank could not map its peers' mailboxes: {errs[0]}")
            elif err is not None:
                self.close()
                raise RuntimeError(f"MailboxAllReduce: {err}")

    def exchange(self, packed: torch.Tensor, out: torch.Tensor, stream: Optional[int] = None) -> None:
        """out[i] = sum over ranks (in rank order) of packed[i]; both contiguous fp32 device buffers of n_floats; asynchronous on
        `stream` (raw HIP stream handle; default: torch's current stream).  out may not alias packed."""
        for name, t in (("packed", packed), ("out", out)):
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != self.n_floats:
                raise ValueError(f"MailboxAllReduce.exchange({name}): expected a contiguous float32 tensor of {self.n_floats} elements on {self.device}")
        with torch.cuda.device(self.device):             # the launch goes to the CURRENT device: make it the mailbox's
            if stream is None:
                stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._lib.trk_mailbox_exchange(self._h, packed.data_ptr(), out.data_ptr(), stream), "trk_mailbox_exchange")

    def send(self, packed: torch.Tensor, stream: Optional[int] = None) -> None:
        """The first half of `exchange`: this rank's row into every mailbox + the flag.  Never waits."""
        if packed.device != self.device or packed.dtype != torch.float32 or not packed.is_contiguous() or packed.numel() != self.n_floats:
            raise ValueError(f"MailboxAllReduce.send(packed): expected a contiguous float32 tensor of {self.n_floats} elements on {self.device}")
        with torch.cuda.device(self.device):
            if stream is None:
                stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._lib.trk_mailbox_send(self._h, packed.data_ptr(), stream), "trk_mailbox_send")

    def recv(self, out: torch.Tensor, stream: Optional[int] = None) -> None:
        """The second half: waits for the rows of the oldest send not yet received and writes their sum (rank order) to out.  Must
        follow its send in stream order (the same stream, or a stream that waits for it); send k + a may precede recv k only for
        a <= (n_slots - 2) / 2 (the library refuses a send beyond that).  A wait that times out writes NaN to all of `out` and counts in
        `status()[1]` (sticky) -- `healthy()` is the cheap check."""
        if out.device != self.device or out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != self.n_floats:
            raise ValueError(f"MailboxAllReduce.recv(out): expected a contiguous float32 tensor of {self.n_floats} elements on {self.device}")
        with torch.cuda.device(self.device):
            if stream is None:
                stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._lib.trk_mailbox_recv(self._h, out.data_ptr(), stream), "trk_mailbox_recv")

    def healthy(self) -> bool:
        """No receive of this mailbox has timed out so far (synchronises with the device)."""
        return self.status()[1] == 0

    def status(self):
        """(exchanges issued, time-outs seen, allocation kind); synchronises with the device."""
        n, t, k = C.c_int64(), C.c_int64(), C.c_int32()
        with torch.cuda.device(self.device):
            self._check(self._lib.trk_mailbox_status(self._h, C.byref(n), C.byref(t), C.byref(k)), "trk_mailbox_status")
        return int(n.value), int(t.value), self.ALLOC_KINDS[k.value] if 0 <= k.value < 3 else "?"

    def validate(self, rounds: int = 6) -> bool:
        """Known-answer exchanges (more rounds than slots, so that slot reuse is exercised): every rank contributes
        (rank + 1) * (i + round) and must read world (world + 1) / 2 * (i + round).  Collective: True only if EVERY rank saw the
        right sums and no time-out."""
        idx = torch.arange(self.n_floats, device=self.device, dtype=torch.float32)
        out = torch.empty(self.n_floats, device=self.device, dtype=torch.float32)
        ok = True
        for r in range(rounds):
            packed = (idx + float(r)) * float(self.rank + 1)
            self.exchange(packed, out)
            expect = (idx + float(r)) * float(self.world * (self.world + 1) // 2)
            ok = ok and bool(torch.equal(out, expect))          # the comparison synchronises
            if not ok:                                           # (a time-out costs TRK_MAILBOX_TIMEOUT_S: do not pay it `rounds` times)
                break
        ok = ok and self.status()[1] == 0
        if self.world > 1:
            flag = torch.tensor([1.0 if ok else 0.0], device=self.device if dist.get_backend(self.group) != "gloo" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ok = bool(flag.item() > 0.5)
        return ok

    def close(self) -> None:
        h, self._h = self._h, C.c_void_p()
        if h:
            self._lib.trk_mailbox_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedRollout:
    """A batch-sharded planner's evaluation + exchange as ONE object (what is sharded is the batch of
    `PlanningTask._compute_collision_or_cost`, tasks.py:206-230): this rank's block of whole trajectories is evaluated by a pre-bound
    fused rollout (`ops.RolloutPlan` / `ops.RolloutGpPlan`), and `exchange()` makes the packed sums

        [ sum cost | sum_b cost(b, h) (H) | sum_b d cost / d q (b, h, d) (H D) ]

    of ALL ranks available on every rank -- through the peer-to-peer mailbox when it validates on every rank (default), else through an
    all-reduce of the process group (RCCL on the GPU box, gloo in tests).  Per-sample outputs (cost, gradient, link positions) stay
    sharded.  Everything is asynchronous on torch's current stream (or `stream`):

        sh = ShardedRollout(plan)                 # collective: every rank constructs it
        for it in range(n_iters):
            sh.launch()                           # plan.cost / plan.gq / plan.link_pos of THIS rank's trajectories
            sh.send()                             # pack + (mailbox) store into every peer -- never waits
            ...                                   # more work of this iteration
            total = sh.recv()                     # (1 + H + H D,) fp32, the sums over all ranks; same bits on every rank (mailbox)

    `exchange()` = `send()` + `recv()`.  With one rank (no process group) the sums are this rank's own."""

    def __init__(self, plan, traj_cost: Optional[torch.Tensor] = None, group: Optional[dist.ProcessGroup] = None, exchange: str = "auto"):
        from . import ops
        if exchange not in ("auto", "p2p", "allreduce"):
            raise ValueError("ShardedRollout: exchange must be 'auto', 'p2p' or 'allreduce'")
        self.plan, self.group, self.device = plan, group, plan.device
        have_group = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if have_group else 1
        self.rank = dist.get_rank(group) if have_group else 0
        B, H = plan.B, plan.H
        self.block_sums = torch.zeros(ops.n_blocks(B * H), device=self.device, dtype=torch.float32)
        self._packer = ops.PackedSums(plan, self.block_sums, traj_cost)
        self.size = self._packer.size
        self.packed = torch.zeros(self.size, device=self.device, dtype=torch.float32)
        self.total = torch.zeros(self.size, device=self.device, dtype=torch.float32)
        self.mailbox, self.mailbox_note = None, None
        if self.world > 1 and exchange in ("auto", "p2p"):
            try:
                mb = MailboxAllReduce(self.device, self.size, n_slots=4, group=group)
                if mb.validate():
                    self.mailbox = mb
                else:
                    self.mailbox_note = "validation failed: " + repr(mb.status())
                    mb.close()
            except Exception as e:          # noqa: BLE001 -- any failure to share memory means: use the collective
                self.mailbox_note = f"{type(e).__name__}: {e}"
            if self.mailbox is None and exchange == "p2p":
                raise RuntimeError(f"ShardedRollout(exchange='p2p'): the mailbox is not usable: {self.mailbox_note}")
        self._sent = False
        self.check_every = int(os.environ.get("TRK_MAILBOX_CHECK_EVERY", "256"))      # receives between two health checks (0: only in close())
        self._recvs = 0

    def _check_mailbox(self) -> None:
        """A timed-out receive wrote NaN sums: raise instead of letting a planner iterate on them.  Synchronises with the device."""
        if self.mailbox is not None and not self.mailbox.healthy():
            n, t, kind = self.mailbox.status()
            raise RuntimeError(f"ShardedRollout: {t} mailbox receive(s) of {n} exchanges timed out on rank {self.rank} (a peer is dead or more than "
                               f"TRK_MAILBOX_TIMEOUT_S late); the sums of those receives are NaN.  Rebuild with exchange='allreduce' or restart the job.")

    def launch(self, stream: Optional[int] = None) -> None:
        self.plan.launch(self.block_sums.data_ptr(), stream)

    def send(self, stream: Optional[int] = None) -> None:
        """pack this rank's sums of the latest `launch()`; mailbox mode: store them into every peer's mailbox.  Never waits."""
        if self._sent:
            raise RuntimeError("ShardedRollout.send: the previous exchange has not been received yet (send / recv alternate)")
        self._packer.pack(self.packed, stream)
        if self.mailbox is not None:
            self.mailbox.send(self.packed, stream)
        self._sent = True

    def recv(self, stream: Optional[int] = None) -> torch.Tensor:
        """the sums over all ranks of the latest `send()` -> `self.total` (returned; valid once the stream has run)"""
        if not self._sent:
            raise RuntimeError("ShardedRollout.recv: nothing has been sent")
        self._sent = False
        if self.mailbox is not None:
            self.mailbox.recv(self.total, stream)
            self._recvs += 1
            if self.check_every > 0 and self._recvs % self.check_every == 0:
                self._check_mailbox()
            return self.total
        cur = torch.cuda.current_stream(self.device)
        if stream is not None and stream != cur.cuda_stream:
            raise ValueError("ShardedRollout.recv: the all-reduce path runs on torch's current stream")
        self.total.copy_(self.packed)
        if self.world > 1:
            dist.all_reduce(self.total, group=self.group)
        return self.total

    def exchange(self, stream: Optional[int] = None) -> torch.Tensor:
        self.send(stream)
        return self.recv(stream)

    def close(self) -> None:
        """Releases the mailbox; raises if any of its receives had timed out (their sums were NaN)."""
        if self.mailbox is not None:
            try:
                self._check_mailbox()
            finally:
                self.mailbox.close()
                self.mailbox = None
