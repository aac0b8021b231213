"""Data-parallel gradient exchange for the translator: one process per GPU, clips sharded by rank, ONE
all-reduce(sum)/world of the flat gradient buffer over RCCL (backend "nccl" on ROCm) / xGMI.

The reference gets this from Lightning's DDP reducer (HOI/scripts/multitask/run.py:41-50, strategy="ddp").
Here the backward of the encoder already emits every gradient as a view of one flat fp32 buffer
(functional._GradPacker), so the exchange is a single collective on that buffer (2.77 MB for the 3-task TTM
translator) instead of per-parameter buckets; the few remaining gradients (task head) are coalesced into one
more small buffer. Works with gloo on CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch
import torch.distributed as dist


# ---- which collective runs the exchanges ------------------------------------------------------------------------------------------------
# Default: torch.distributed (the process group Lightning's DDP would hand over; backend "nccl" = RCCL). use_egx_comm(EgxComm) routes EVERY
# exchange of this module through egx_allreduce instead: plain RCCL calls on torch streams this module owns, no ProcessGroupNCCL work
# objects and no watchdog thread. That is what a step captured into a hipGraph needs (round 6, VERDICT r5 item 5: PyTorch's watchdog
# queried an event of the capturing stream and aborted one profile run in six with hipErrorCapturedEvent; a retry hid it).
_comm = None
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def use_egx_comm(comm) -> None:
    """comm: an EgxComm (or None to return to torch.distributed). Every rank must make the same choice."""
    global _comm
    _comm = comm


def _world(group=None) -> int:
    if _comm is not None:
        return _comm.world
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _active(group=None, force: bool = False) -> bool:
    if _comm is not None:
        return _comm.world > 1 or force
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)


class _SideWork:
    """Handle of a collective issued on this module's side stream: wait() makes the current stream wait for it (an event, no host sync;
    under capture: the join of a parallel branch of the graph)."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)


def _all_reduce(buf: torch.Tensor, average: bool, group=None, async_op: bool = False):
    """In-place all-reduce of `buf`; returns (work | None, averaged_by_the_collective)."""
    if _comm is not None:
        if not async_op:
            _comm.allreduce_(buf, average)
            return None, True
        dev = buf.device.index
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(device=buf.device)
        side.wait_stream(torch.cuda.current_stream(buf.device))      # behind the kernels that completed `buf`
        with torch.cuda.stream(side):
            _comm.allreduce_(buf, average)
        return _SideWork(side), True
    fused_avg = average and dist.get_backend(group) == "nccl"       # RCCL averages inside the collective (ncclAvg); gloo (CPU tests) needs the explicit scale
    op = dist.ReduceOp.AVG if fused_avg else dist.ReduceOp.SUM
    work = dist.all_reduce(buf, op=op, group=group, async_op=async_op)
    return (work if async_op else None), fused_avg


def shard_batch(tensors: Iterable[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Rank r takes clips [r*B/n, (r+1)*B/n) of every feature tensor (DistributedSampler equivalent,
    HOI/tasks/multitask/video_task.py:631)."""
    out = []
    for t in tensors:
        B = t.shape[0]
        if B % world:
            raise ValueError(f"batch {B} not divisible by world size {world}")
        n = B // world
        out.append(t[rank * n:(rank + 1) * n])
    return out


def _flat_groups(params: Iterable[torch.nn.Parameter]):
    """Group gradients by the storage they share (the encoder's backward carves all of them out of one flat buffer);
    gradients that own their storage alone are returned separately and coalesced by the caller."""
    from .functional import flat_storage_view
    count: Dict[int, int] = {}
    grads = [p.grad for p in params if p.grad is not None]
    for g in grads:
        k = g.untyped_storage().data_ptr()
        count[k] = count.get(k, 0) + 1
    bases: Dict[int, torch.Tensor] = {}
    loose: List[torch.Tensor] = []
    for g in grads:
        k = g.untyped_storage().data_ptr()
        if count[k] > 1 and g.is_contiguous():
            if k not in bases:
                bases[k] = flat_storage_view(g)
        else:
            loose.append(g)
    return list(bases.values()), loose


@torch.no_grad()
def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, average: bool = True, force: bool = False) -> int:
    """All-reduce every .grad in place; returns the number of collectives issued. `force` issues the collectives even in
    a one-rank group (used to exercise the RCCL path on a single GPU)."""
    if not _active(group, force):
        return 0
    world = _world(group)
    params = [p for p in params if p.grad is not None]
    flats, loose = _flat_groups(params)
    n = 0
    for buf in flats:
        _, fused_avg = _all_reduce(buf, average, group)
        if average and not fused_avg:
            buf.mul_(1.0 / world)
        n += 1
    if loose:
        flat = torch.cat([g.reshape(-1) for g in loose])
        _, fused_avg = _all_reduce(flat, average, group)
        if average and not fused_avg:
            flat.mul_(1.0 / world)
        off = 0
        for g in loose:
            k = g.numel()
            g.copy_(flat[off:off + k].view_as(g))
            off += k
        n += 1
    return n


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Replicate rank-0 weights (what DDP does at construction)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


@torch.no_grad()
def allreduce_gradients_overlapped(finish_backward, params: Iterable[torch.nn.Parameter] = None, group=None,
                                   average: bool = True, force: bool = False) -> int:
    """Gradient exchange overlapped with the tail of the backward (BASELINE.json north_star). Use with a model whose
    `egx_defer_small` is set: its backward then stops before the grouped small weight gradients (dW_proj, dW_in, dW_o),
    which sit FIRST in the flat gradient buffer. This call starts the all-reduce of everything behind them on RCCL's
    stream, runs `finish_backward()` (functional.run_deferred, or the replay of a graph that captured it) on the compute
    stream meanwhile, then all-reduces the late region, and finally every gradient of `params` that does NOT live in
    that flat buffer (task heads run through PoolHeadFn / LinearFn, decoder parameters, a second encoder call) the way
    allreduce_gradients does. The recorded flat buffer must be the storage of gradients of `params`: a layout left
    behind by some other backward raises instead of silently exchanging the wrong buffer. Returns the number of
    collectives issued."""
    from . import functional as F_egx
    lay = F_egx.last_grad_layout
    flat, late = lay.get("flat"), int(lay.get("late_floats", 0))
    world = _world(group)
    if not _active(group, force):
        finish_backward()
        return 0
    params = [p for p in params if p.grad is not None] if params is not None else None
    if flat is None:
        finish_backward()
        return allreduce_gradients(params, group, average, force) if params is not None else 0
    rest = []
    if params is not None:
        key = flat.untyped_storage().data_ptr()
        inside = [p for p in params if p.grad.untyped_storage().data_ptr() == key]
        rest = [p for p in params if p.grad.untyped_storage().data_ptr() != key]
        if not inside:
            raise RuntimeError("allreduce_gradients_overlapped: the recorded flat gradient buffer does not hold any gradient of "
                               "`params` (another backward ran in between?)")
    early_buf, late_buf = flat[late:], flat[:late]
    fused_avg = True
    work = None
    if early_buf.numel():
        work, fused_avg = _all_reduce(early_buf, average, group, async_op=True)
    finish_backward()                       # overlaps the collective
    n = 1 if work is not None else 0
    if late_buf.numel():
        _, fused_avg = _all_reduce(late_buf, average, group)
        n += 1
    if work is not None:
        work.wait()
    if average and not fused_avg:
        flat.mul_(1.0 / world)
    if rest:
        n += allreduce_gradients(rest, group, average, force)
    return n


class BucketedExchange:
    """Gradient exchange overlapped with the backward, bucket by bucket (the wide bf16 path: BASELINE.json configs[3] / [4], the
    8-GPU EgoT2-g configuration; the reference gets it from DDP's bucketed reducer, HOI/scripts/multitask/run.py:41-50):

        with ddp.BucketedExchange(params) as ex:
            loss.backward()
        # every .grad of `params` is now the average over the ranks; ex.collectives were issued

    Inside the block every backward of the library announces a slice of its flat gradient buffer as soon as the kernels that
    complete it are enqueued (functional.bucket_hook): the wide encoder one slice per layer, last layer first — the buffer is
    laid out in that order — the sequence decoder its whole buffer (it runs before the encoder's backward). Each announcement
    starts an asynchronous all-reduce of that slice; RCCL orders it behind the kernels enqueued so far and runs it beside the
    ones enqueued afterwards. Leaving the block waits for the collectives and exchanges the gradients that were not announced
    (task heads computed by torch) the way allreduce_gradients does. `force` runs the collectives in a one-rank group.

    Precondition (checked on entry): no parameter holds a .grad yet — the block must wrap a backward that starts from
    `zero_grad(set_to_none=True)`. Only PARAMETER gradients live in the announced buffer: activation gradients the backward
    returns (d(feature) of a trainable upstream head) are separate tensors and are never exchanged."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, average: bool = True, force: bool = False):
        self.params, self.group, self.average, self.force = list(params), group, average, force
        self.collectives = 0
        self._work, self._covered = [], []
        self.active = _active(group, force)

    def _on_bucket(self, flat: torch.Tensor, lo: int, hi: int):
        if hi <= lo:
            return
        buf = flat[lo:hi]
        work, fused_avg = _all_reduce(buf, self.average, self.group, async_op=True)
        self._work.append((work, buf, fused_avg))
        self._covered.append((flat.untyped_storage().data_ptr(), flat.data_ptr() + 4 * lo, flat.data_ptr() + 4 * hi))
        self.collectives += 1

    def __enter__(self):
        from . import functional as F_egx
        if self.active:
            if F_egx.bucket_hook is not None:
                raise RuntimeError("BucketedExchange: another exchange is active")
            # Precondition: the wrapped backward must CREATE every .grad (as a view of the announced flat buffer). With a
            # .grad already present autograd accumulates `p.grad += view` on the compute stream while the asynchronous
            # all-reduce rewrites that view, and the parameter would be exchanged a second time on exit.
            stale = sum(1 for p in self.params if p.grad is not None)
            if stale:
                raise RuntimeError(f"BucketedExchange: {stale} parameter(s) already hold a .grad; call "
                                   "optimizer.zero_grad(set_to_none=True) first. For gradient accumulation wrap only the LAST "
                                   "micro-batch's backward in allreduce_gradients(...) after it instead of this exchange")
            F_egx.bucket_hook = self._on_bucket
        return self

    @torch.no_grad()
    def __exit__(self, exc_type, exc, tb):
        from . import functional as F_egx
        if not self.active:
            return False
        F_egx.bucket_hook = None
        world = _world(self.group)
        for work, buf, fused_avg in self._work:
            work.wait()
            if self.average and not fused_avg:
                buf.mul_(1.0 / world)
        if exc_type is None:
            def covered(g):
                a = g.data_ptr()
                return any(lo <= a < hi for _, lo, hi in self._covered)
            rest = [p for p in self.params if p.grad is not None and not covered(p.grad)]
            if rest:
                self.collectives += allreduce_gradients(rest, self.group, self.average, self.force)
        self._work = []
        return False


class EgxComm:
    """RCCL communicator under the C ABI (include/egot2x.h: egx_comm_* / egx_allreduce) for harnesses that have no torch process
    group. One per process, on the current device. The 128-byte id is drawn by rank 0 and carried to the others by `bcast`
    (any callable bytes -> bytes that returns rank 0's argument on every rank: an MPI bcast, a file, a torch.distributed
    object broadcast). `allreduce_gradients(params)` is the same exchange as the module-level function: the flat gradient
    buffer(s) of the backward, summed and divided by the rank count, asynchronous on the current stream."""

    def __init__(self, rank: int, world: int, bcast=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        buf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(self._lib.egx_comm_unique_id(buf))
        idb = bytes(buf)
        if self.world > 1:
            if bcast is None:
                raise ValueError("EgxComm: world > 1 needs a `bcast` callable to carry rank 0's id to the other ranks")
            idb = bcast(idb)
        h = C.c_void_p()
        _lib.check(self._lib.egx_comm_create(C.c_char_p(idb), self.rank, self.world, C.byref(h)))
        self._h = h
        self.device = torch.cuda.current_device()       # the communicator is bound to the device that was current here

    @property
    def size(self) -> int:
        return int(self._lib.egx_comm_size(self._h))

    def allreduce_(self, t: torch.Tensor, average: bool = True) -> torch.Tensor:
        from . import _lib
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.float32, torch.bfloat16):
            raise _lib.EgxError("EgxComm.allreduce_: contiguous fp32 / bf16 CUDA tensor expected")
        if t.device.index != self.device:
            raise _lib.EgxError(f"EgxComm.allreduce_: tensor on cuda:{t.device.index}, communicator created on cuda:{self.device}")
        if self._h is None:
            raise _lib.EgxError("EgxComm.allreduce_: communicator is closed")
        _lib.check(self._lib.egx_allreduce(self._h, t.data_ptr(), t.numel(), int(t.dtype == torch.bfloat16), int(average),
                                           torch.cuda.current_stream(t.device).cuda_stream))
        return t

    @torch.no_grad()
    def allreduce_gradients(self, params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
        flats, singles = _flat_groups([p for p in params if p.grad is not None])
        n = 0
        for f in flats:
            self.allreduce_(f, average)
            n += 1
        if singles:
            coalesced = torch.cat([g.reshape(-1) for g in singles])
            self.allreduce_(coalesced, average)
            off = 0
            for g in singles:
                g.copy_(coalesced[off:off + g.numel()].view_as(g))
                off += g.numel()
            n += 1
        return n

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.egx_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        import sys
        if sys is None or sys.is_finalizing():      # interpreter shutdown: RCCL / HIP may already be torn down, leave the communicator to the process exit
            return
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass
