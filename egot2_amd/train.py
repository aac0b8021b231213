"""Training-step pieces around the translator: the TTM criterion and the optimizer.

`CrossEntropyLoss` mirrors `nn.CrossEntropyLoss(weight=...)` as built at HHI/tasks/ttm/video_task_2loader.py:21-22
(same constructor argument and `weight` buffer name, so Lightning checkpoints carrying `criterion.weight` load).
`FusedAdam` mirrors `torch.optim.Adam` / `AdamW` as configured at HHI/tasks/ttm/video_task_2loader.py:62-64 and
HOI/tasks/multitask/video_task.py:624-626: the translator's gradients already arrive as views of ONE flat buffer
(functional._GradPacker), so the parameters are re-pointed into a flat buffer with the same layout and the whole
update is a single launch per buffer, with the step count in device memory (hipGraph-replayable).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_egx
from ._lib import check, ptr


class CrossEntropyLoss(nn.Module):
    def __init__(self, weight: Optional[torch.Tensor] = None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else weight.detach().clone().float())

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return F_egx.weighted_cross_entropy(input, target, self.weight)


class _Bucket:
    __slots__ = ("sig", "param", "exp_avg", "exp_avg_sq", "numel", "members")


class FusedAdam(torch.optim.Optimizer):
    """Adam (adamw=False) / AdamW (adamw=True) with torch.optim semantics, one launch per flat gradient buffer."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, adamw: bool = False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adamw=adamw))
        self._buckets: Dict[tuple, _Bucket] = {}
        self._step_dev: Optional[torch.Tensor] = None
        self._resume_step = 0          # step count restored by load_state_dict before the device counter exists

    def _make_bucket(self, sig, base: torch.Tensor, members) -> _Bucket:
        b = _Bucket()
        b.sig = sig
        b.numel = base.numel()
        if len(members) == 1 and members[0][1] == 0 and members[0][0].numel() == base.numel() and members[0][0].is_contiguous():
            b.param = members[0][0].data.view(-1)          # stand-alone tensor: update it in place where it lives
        else:
            b.param = torch.zeros_like(base)
            for p, off in members:                          # move the parameters behind the gradient layout
                n = p.numel()
                b.param[off:off + n].copy_(p.data.reshape(-1))
                p.data = b.param[off:off + n].view(p.shape)
        b.exp_avg = torch.zeros_like(b.param)
        b.exp_avg_sq = torch.zeros_like(b.param)
        b.members = list(members)
        self._adopt_state(b)
        return b

    def _adopt_state(self, b: _Bucket):
        """Make self.state[p] views of the bucket's moment buffers. Moments that are already there (restored by
        load_state_dict, or carried over from a bucket that had to be rebuilt) are copied in first, so a resumed run
        continues from the saved exp_avg / exp_avg_sq instead of restarting them from zero."""
        for p, off in b.members:
            n = p.numel()
            old = self.state.get(p)
            for key, buf in (("exp_avg", b.exp_avg), ("exp_avg_sq", b.exp_avg_sq)):
                view = buf[off:off + n]
                if old is not None and torch.is_tensor(old.get(key)) and old[key].numel() == n \
                        and old[key].data_ptr() != view.data_ptr():
                    view.copy_(old[key].reshape(-1).to(view.device, view.dtype))
            self.state[p] = {"step": self._step_dev, "exp_avg": b.exp_avg[off:off + n].view(p.shape),
                             "exp_avg_sq": b.exp_avg_sq[off:off + n].view(p.shape)}

    def load_state_dict(self, state_dict):
        """torch's Optimizer.load_state_dict replaces self.state with fresh copies; the real moments and the step count
        live in the bucket buffers / the device counter, so push the loaded values back into them (buckets that do not
        exist yet pick them up in _make_bucket). Mirrors the optimizer resume of the reference's Lightning trainer
        (`trainer.fit(ckpt_path=...)`, HOI/scripts/pnr/train.py:57)."""
        super().load_state_dict(state_dict)
        steps = [st["step"] for st in self.state.values() if st.get("step") is not None]
        step = max((int(s.item()) if torch.is_tensor(s) else int(s)) for s in steps) if steps else 0
        self._resume_step = step
        if self._step_dev is not None:
            self._step_dev.fill_(step)
        for b in self._buckets.values():
            for p, off in b.members:        # a parameter WITHOUT loaded state starts from zero moments again (a state_dict taken before the first step)
                if p not in self.state:
                    b.exp_avg[off:off + p.numel()].zero_()
                    b.exp_avg_sq[off:off + p.numel()].zero_()
            self._adopt_state(b)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        bumped = False
        for gi, group in enumerate(self.param_groups):
            by_base: Dict[int, Tuple[torch.Tensor, List]] = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous():
                    raise ValueError("FusedAdam needs contiguous fp32 gradients on the GPU")
                key = g.untyped_storage().data_ptr()        # the flat buffer this gradient was carved out of: one view per buffer,
                ent = by_base.get(key)                       # not one per parameter (25 tensor constructions per step on the TTM translator)
                if ent is None:
                    ent = by_base[key] = (F_egx.flat_storage_view(g), [])
                ent[1].append((p, g.storage_offset()))
            for base, members in by_base.values():
                if self._step_dev is None:
                    self._step_dev = torch.full((), self._resume_step, dtype=torch.int64, device=base.device)
                if not bumped:
                    check(lib.egx_counter_add(ptr(self._step_dev), 1, stream))
                    bumped = True
                sig = (gi,) + tuple((id(p), off) for p, off in members) + (base.numel(),)
                b = self._buckets.get(sig)
                if b is not None and any(p.data_ptr() != b.param.data_ptr() + 4 * off for p, off in members):
                    b = None                                 # somebody re-allocated a parameter: rebuild the bucket
                if b is None:
                    b = self._buckets[sig] = self._make_bucket(sig, base.reshape(-1), members)
                gflat = base.reshape(-1)
                F_egx.note_weights_changed()       # (raw-pointer update: no version counter moves; packed-weight caches re-pack)
                check(lib.egx_adam_step(ptr(b.param), ptr(gflat), ptr(b.exp_avg), ptr(b.exp_avg_sq), b.numel,
                                        ptr(self._step_dev), group["lr"], group["betas"][0], group["betas"][1],
                                        group["eps"], group["weight_decay"], int(group["adamw"]), 1.0, stream))
        return loss


class GraphedStep:
    """One training step — forward, loss, backward and (optionally) the optimizer update — captured ONCE as a hipGraph and replayed.

    The library only enqueues kernels on the current stream and keeps what changes from step to step in device memory (the dropout seed after
    `model.enable_device_seed()`, FusedAdam's step count), so a replay is exact training work with fresh masks; what the capture removes is the
    host side of the step (Python, the autograd engine's worker thread, a dozen launches: 0.54 - 0.90 ms per step on the bench configuration with
    FusedAdam against 0.41 ms of GPU time). This is the
    loop `bench.py` times, packaged for a training script:

        step = GraphedStep(lambda f, y: criterion(model.forward_features(*f), y), example_inputs=(feats, target),
                           params=model.parameters(), optimizer=FusedAdam(model.parameters(), lr=1e-4))
        for feats, target in loader:
            loss = step(feats, target)          # copies the batch into the captured buffers, replays, returns the loss tensor (device)

    `loss_fn(*inputs)` must be a pure function of its (possibly nested list / tuple of) tensor arguments with FIXED shapes and dtypes: a batch of
    another shape needs another GraphedStep (the reference's length-bucketed TTM sampler yields a handful of shapes — keep one per shape).
    Gradients stay in `.grad` after every replay (`optimizer=None`: update them yourself outside the graph)."""

    def __init__(self, loss_fn, example_inputs, params, optimizer=None, warmup: int = 3):
        self.loss_fn, self.optimizer = loss_fn, optimizer
        self.params = [p for p in params if p.requires_grad]
        self.static = self._clone_tree(example_inputs)
        self._one = None
        dev = self.params[0].device
        # The warm-up runs REAL steps on `example_inputs` (allocator, lazy initialisation, FusedAdam re-pointing the parameters into its flat
        # buffers: all of that has to happen before the capture). With an optimizer they would leave `warmup` updates from the example batch
        # behind: parameters, optimizer moments / step count and the device-resident dropout seeds are snapshotted here and
        # restored afterwards, so the first replay starts exactly where the caller's model and optimizer were.
        snap_p = [p.detach().clone() for p in self.params]
        snap_o = None
        if optimizer is not None:
            import copy
            snap_o = copy.deepcopy(optimizer.state_dict())
        seeds = [m._egx_seed_dev for m in self._seed_owners(loss_fn)]
        snap_s = [t.clone() for t in seeds]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._step()
            with torch.no_grad():
                for p, v in zip(self.params, snap_p):
                    p.copy_(v)
                for t, v in zip(seeds, snap_s):
                    t.copy_(v)
            if optimizer is not None:
                optimizer.load_state_dict(snap_o)
            F_egx.note_weights_changed()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._step()
        torch.cuda.synchronize(dev)

    @staticmethod
    def _seed_owners(loss_fn):
        """Modules with a device-resident dropout seed that `loss_fn` closes over (model.enable_device_seed())."""
        found = []
        cells = getattr(loss_fn, "__closure__", None) or ()
        objs = [c.cell_contents for c in cells if c is not None]
        objs += [getattr(loss_fn, "__self__", None)]
        for o in objs:
            mods = o.modules() if isinstance(o, nn.Module) else ()
            for m in mods:
                if isinstance(getattr(m, "_egx_seed_dev", None), torch.Tensor) and all(m is not f for f in found):
                    found.append(m)
        return found

    def _clone_tree(self, t):
        if isinstance(t, torch.Tensor):
            return t.detach().clone()
        if isinstance(t, (list, tuple)):
            return type(t)(self._clone_tree(x) for x in t)
        raise TypeError(f"GraphedStep: inputs must be tensors or nested lists / tuples of tensors, got {type(t).__name__}")

    def _copy_tree(self, dst, src):
        if isinstance(dst, torch.Tensor):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"GraphedStep: input of shape {tuple(src.shape)} / {src.dtype} where the step was captured with "
                                 f"{tuple(dst.shape)} / {dst.dtype} (one GraphedStep per batch shape)")
            dst.copy_(src, non_blocking=True)
        else:
            if len(dst) != len(src):
                raise ValueError("GraphedStep: input structure differs from the captured one")
            for d, s in zip(dst, src):
                self._copy_tree(d, s)

    def _step(self):
        for p in self.params:
            p.grad = None
        loss = self.loss_fn(*self.static)
        if self._one is None:
            self._one = torch.ones_like(loss)
        loss.backward(gradient=self._one)
        if self.optimizer is not None:
            self.optimizer.step()
        return loss.detach()

    def __call__(self, *inputs):
        self._copy_tree(self.static, type(self.static)(inputs) if isinstance(self.static, (list, tuple)) else inputs[0])
        self.graph.replay()
        if self.optimizer is not None:
            F_egx.note_weights_changed()           # the replayed update wrote the parameters behind torch's back
        return self.loss
