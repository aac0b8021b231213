"""torch.autograd bridges onto the C ABI of libegot2x.so.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic of the translator runs in the
hand-written HIP kernels. There is no fallback: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (Ce, Config, Head, HeadGrads, Layer, LayerGrads, Segment, SegmentGrads, check, ptr, _LAYER_FIELDS,
                   EGX_F32, EGX_BF16, EGX_F32_SPLIT, EGX_IMPL_AUTO, EGX_IMPL_GENERIC, EGX_IMPL_FUSED, EGX_IMPL_WIDE, EGX_IMPL_TILED)

# "f32s": fp32 operands split exactly into three bf16 parts, six bf16 MFMAs per K-block (fp32-grade results, fused d = 128
# kernels only; elsewhere it computes as "f32")
COMPUTE = {"f32": EGX_F32, "fp32": EGX_F32, "float32": EGX_F32, "bf16": EGX_BF16, "bfloat16": EGX_BF16, "f32s": EGX_F32_SPLIT}
IMPL = {"auto": EGX_IMPL_AUTO, "generic": EGX_IMPL_GENERIC, "fused": EGX_IMPL_FUSED, "wide": EGX_IMPL_WIDE, "tiled": EGX_IMPL_TILED}


@dataclass
class SegmentSpec:
    T: int
    d_in: int
    has_proj: bool
    add_row: Optional[int] = None      # row of the (1, K, d) task-embedding table, or None
    pos_row0: Optional[int] = None     # first row of the positional table used by this segment, or None
    pool: int = 1                      # > 1: the feature tensor holds T * pool frames, token t = mean of frames [t pool, (t+1) pool)


_weights_epoch = [0]


def note_weights_changed():
    """Tell the packed-weight caches (WeightCache) that parameters were written behind torch's back (raw-pointer kernels: FusedAdam, a
    replayed hipGraph that contains an optimizer update). In-place torch ops on the parameters are seen through their version counters."""
    _weights_epoch[0] += 1


class WeightCache:
    """Persistent device buffer for the MFMA-fragment-packed weight copies of the per-clip / tiled kernels (egx_config.weight_cache).
    A forward whose weights did not change since the forward that filled it skips the packing launch (6 us of the 0.38 ms TTM step).
    "Did not change" is decided on the host: storage address + version counter of every packed weight, the compute mode, the FFN keep-scale
    and note_weights_changed()'s epoch. Writes that bypass all of those (`p.data.add_()` bumps no counter) need invalidate().
    frozen=True is the caller's promise that the weights do not change while it is set (inference, a forward + backward benchmark): only
    then is the packing launch also left out of a CAPTURED step — a replay cannot re-check anything."""

    def __init__(self, frozen: bool = False):
        self.buf: Optional[torch.Tensor] = None
        self.sig = None
        self.frozen = bool(frozen)
        self.hits = 0
        self.packs = 0

    def invalidate(self):
        self.sig = None


@dataclass
class EncoderSpec:
    d_model: int
    n_heads: int
    d_ff: int
    n_layers: int
    segments: List[SegmentSpec] = field(default_factory=list)
    ln_eps: float = 1e-5
    compute: str = "f32"
    impl: str = "auto"
    p_drop: float = 0.0
    p_pos: float = 0.0
    p_feat: float = 0.0
    training: bool = False
    seed: int = 0
    seed_ptr: int = 0      # device address of a uint64 seed (hipGraph-replayable dropout), 0 = use `seed`
    head_n_out: int = 0    # > 0: pooled head (mean -> LN -> Linear) evaluated with the encoder; output = logits
    advance_seed: int = 0        # with seed_ptr: 1 = the training forward advances the device seed in its first kernel; 2 = the backward
                                 # advances it behind its last reader (egx_config.advance_seed)
    defer_small: bool = False    # backward: leave the grouped small weight gradients to run_deferred() (all-reduce overlap)
    deterministic: bool = False  # backward: fixed-order reductions instead of fp32 atomics (bit-identical gradients run to run)
    out_tokens: int = 0          # > 0: return (and take the gradient of) only the first `out_tokens` tokens of every clip
    wcache: Optional[WeightCache] = None     # persistent packed-weight cache (None: packed into `saved` every forward)
    token_ce: int = 0            # > 0 (classes): the forward also evaluates a per-token classifier + weighted cross entropy on the returned tokens
                                 # (egx_token_ce, the ASD task's lossAV); use encoder_token_ce(), which falls back where the kernels do not fuse it
    ce: bool = False             # with head_n_out: the forward also evaluates the weighted cross entropy of the logits (egx_ce); EncoderFn then
                                 # takes (target, class_weight | None) behind the head parameters and returns (logits, loss)

    def config(self) -> Config:
        return Config(self.d_model, self.n_heads, self.d_ff, self.n_layers, len(self.segments), float(self.ln_eps),
                      COMPUTE[self.compute], IMPL[self.impl], float(self.p_drop), float(self.p_pos), float(self.p_feat),
                      self.seed_ptr or None, int(self.advance_seed) if self.seed_ptr else 0, None, 0, 0,
                      int(bool(self.deterministic)), int(self.out_tokens))


# The library reads its kernel-selection switches (EGX_FFN_CUT, EGX_FFN_SLICES, EGX_SLICE_DROP) from the environment once. Tests and tuning
# tools that flip them inside one process set this (tests/conftest.py): every forward / backward then asks the library to re-read them first.
reload_tuning_each_call = False

_scratch_cache = {}
_last_impl = [EGX_IMPL_AUTO]
_last_slices = [1]


def last_encoder_slices() -> int:
    """Diagnostic: workgroups per clip of the most recent encoder forward (egx_encoder_slices: > 1 = sliced mode of the per-clip
    kernels on a small batch)."""
    return _last_slices[0]


def last_encoder_impl() -> str:
    """Diagnostic: the implementation the most recent encoder forward ran ("fused" | "wide" | "generic" | ...)."""
    return {v: k for k, v in IMPL.items()}.get(_last_impl[0], "auto")

# EGX_POISON=1 (testing aid): every workspace handed to the library is filled with 0xFF bytes (NaN in fp32 and bf16) first,
# so that a kernel reading memory nobody wrote shows up as NaNs instead of passing on whatever the allocator left there
_POISON = bool(int(__import__("os").environ.get("EGX_POISON", "0") or 0))


def _workspace(tag: str, device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only byte buffer per (tag, device). Safe because every call is ordered on the current stream."""
    key = (tag, device.index)
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _scratch_cache[key] = buf
    if _POISON:
        buf.fill_(255)
    return buf


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.EgxError(f"{name} is on {t.device}: the translator runs only on the GPU through libegot2x.so "
                            "(no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.EgxError(f"{name} has dtype {t.dtype}; libegot2x stores activations and weights in fp32")
    return t.contiguous()


def _dev_feat(t: torch.Tensor, name: str) -> torch.Tensor:
    """Backbone features: fp32, or bf16 in the packed layout the wide path's projection GEMM reads."""
    if not t.is_cuda:
        raise _lib.EgxError(f"{name} is on {t.device}: the translator runs only on the GPU through libegot2x.so "
                            "(no CPU fallback)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.EgxError(f"{name} has dtype {t.dtype}; features are fp32 or bf16")
    return t.contiguous()


def _elem_ptr(t: torch.Tensor, row: int, row_elems: int) -> int:
    return t.data_ptr() + 4 * row * row_elems


class _GradPacker:
    """Allocates every requested gradient as a 16-byte-aligned view of one flat buffer. Entries marked `late` (the
    weight gradients the backward finishes last: dW_proj, dW_in, dW_o) are placed first, so that everything behind
    `late_floats` can be all-reduced while they are still being computed (ddp.allreduce_gradients_overlapped)."""

    def __init__(self):
        self.entries = []
        self.total = 0
        self.late_floats = 0

    def add(self, t: Optional[torch.Tensor], want: bool, late: bool = False, rank: int = 0) -> int:
        """`rank`: entries are laid out by ascending rank (stable): the bucketed exchange gives the tensors of the encoder layer the
        backward finishes first the lowest rank, so every layer's gradients are one contiguous slice (self.rank_span)."""
        if t is None or not want:
            self.entries.append(None)
            return -1
        self.entries.append((t.shape, (t.numel() + 3) // 4 * 4, late, rank))
        return len(self.entries) - 1

    _layouts: dict = {}          # entries signature -> (offsets, total, late_floats, rank_span, strides): a model's backward lays out the same buffer every step

    def materialise(self, device, zero: bool = True) -> List[Optional[torch.Tensor]]:
        """zero=False: the caller hands `self.flat` to the library, whose backward zero-fills it (egx_config.zero_buf)."""
        sig = tuple(self.entries)
        lay = _GradPacker._layouts.get(sig)
        if lay is None:
            offs, cur, rank_span, late_floats = {}, 0, {}, 0
            order = sorted((i for i, e in enumerate(self.entries) if e is not None), key=lambda i: self.entries[i][3])
            for want_late in (True, False):
                for i in order:
                    e = self.entries[i]
                    if e[2] == want_late:
                        offs[i] = cur
                        lo, hi = rank_span.get(e[3], (cur, cur))
                        rank_span[e[3]] = (min(lo, cur), cur + e[1])
                        cur += e[1]
                if want_late:
                    late_floats = cur
            strides = {}
            for i, e in enumerate(self.entries):
                if e is not None:
                    st, acc = [], 1
                    for k in reversed(e[0]):
                        st.append(acc)
                        acc *= k
                    strides[i] = tuple(reversed(st))
            if len(_GradPacker._layouts) > 256:
                _GradPacker._layouts.clear()
            lay = _GradPacker._layouts[sig] = (offs, cur, late_floats, rank_span, strides)
        offs, self.total, self.late_floats, rank_span, strides = lay
        self.rank_span = dict(rank_span)
        n = max(self.total, 4)
        flat = torch.zeros(n, dtype=torch.float32, device=device) if zero else torch.empty(n, dtype=torch.float32, device=device)
        if _POISON and not zero:
            flat.fill_(float("nan"))
        self.flat = flat
        # one as_strided per gradient (a slice + a view were two tensor constructions each: 50 per backward of the TTM translator)
        return [None if e is None else flat.as_strided(e[0], strides[i], offs[i]) for i, e in enumerate(self.entries)]


_deferred = []          # closures that finish a staged backward (EncoderSpec.defer_small)
# Bucketed gradient exchange (ddp.BucketedExchange): when set, bucket_hook(flat, lo, hi) is called from inside a backward as
# soon as the kernels that complete flat[lo:hi] are enqueued on the current stream (per encoder layer on the wide path, once
# per buffer elsewhere); the hook starts the all-reduce of that slice, which then overlaps the rest of the backward.
bucket_hook = None
last_grad_layout = {}   # {"flat": flat gradient buffer of the latest encoder backward, "late_floats": size of its late region}


def run_deferred():
    """Launch the grouped small weight gradients that a backward with EncoderSpec.defer_small left out."""
    while _deferred:
        _deferred.pop(0)()


def flat_storage_view(t: torch.Tensor) -> torch.Tensor:
    """1-D view over the WHOLE storage `t` lives in. Gradients handed out by _GradPacker reach `.grad` detached (autograd
    drops `_base`), so the flat buffer behind them is recovered through the storage: every gradient of one backward
    shares it, which is what lets the all-reduce and the optimizer run once per buffer."""
    st = t.untyped_storage()
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(st, 0, (st.nbytes() // t.element_size(),))


class EncoderFn(torch.autograd.Function):
    """tokens(B,S,d) = encoder(token_prep(feats)) — or, with spec.head_n_out > 0, logits(B,n_out) = head(tokens).
    Argument order: spec, task_embed|None, pos_table|None, ln_w, ln_b, feats[n_seg], (proj_w, proj_b) per projecting
    segment, 12 tensors per layer in _LAYER_FIELDS order, then (head_ln_w, head_ln_b, head_W, head_b) when a head
    is requested."""

    @staticmethod
    def forward(ctx, spec: EncoderSpec, task_embed, pos_table, ln_w, ln_b, *rest):
        lib = _lib.load()
        if reload_tuning_each_call:
            lib.egx_tuning_reload()
        nseg = len(spec.segments)
        feats = [_dev_feat(t, f"feats[{i}]") for i, t in enumerate(rest[:nseg])]
        nproj = sum(1 for s in spec.segments if s.has_proj)
        proj = [_dev_f32(t, "projection weight") for t in rest[nseg:nseg + 2 * nproj]]
        nhead = 4 if spec.head_n_out else 0
        tce_in = None
        if spec.token_ce:
            if nhead or spec.ce:
                raise _lib.EgxError("EncoderSpec.token_ce cannot be combined with the pooled head")
            tce_in, rest = rest[-4:], rest[:-4]         # classifier weight, bias | None, target, class weight | None
        ce_target = ce_weight = None
        if spec.ce:
            if not nhead:
                raise _lib.EgxError("EncoderSpec.ce needs the pooled head (head_n_out > 0)")
            ce_target, ce_weight = rest[-2], rest[-1]
            rest = rest[:-2]
        layer_t = [_dev_f32(t, "layer weight") for t in rest[nseg + 2 * nproj:len(rest) - nhead]]
        head_t = [_dev_f32(t, "head parameter") for t in rest[len(rest) - nhead:]] if nhead else []
        assert len(layer_t) == 12 * spec.n_layers, "layer parameter count mismatch"
        ln_w = _dev_f32(ln_w, "ln.weight")
        ln_b = _dev_f32(ln_b, "ln.bias")
        if task_embed is not None:
            task_embed = _dev_f32(task_embed, "task_embed")
        if pos_table is not None:
            pos_table = _dev_f32(pos_table, "positional table")
        d = spec.d_model
        B = feats[0].shape[0]
        device = feats[0].device

        segs = (Segment * nseg)()
        pi = 0
        for i, (ss, f) in enumerate(zip(spec.segments, feats)):
            if f.dim() != 3 or f.shape[0] != B or f.shape[1] != ss.T * max(ss.pool, 1) or f.shape[2] != ss.d_in:
                raise _lib.EgxError(f"feats[{i}] has shape {tuple(f.shape)}, expected ({B}, {ss.T * max(ss.pool, 1)}, {ss.d_in})")
            if not ss.has_proj and (f.dtype == torch.bfloat16 or ss.pool > 1):
                raise _lib.EgxError(f"feats[{i}]: bf16 / frame-pooled features need a projection (an identity segment enters "
                                    "the shared LayerNorm as fp32 rows)")
            segs[i].feat = ptr(f)
            segs[i].T = ss.T
            segs[i].d_in = ss.d_in
            segs[i].feat_bf16 = int(f.dtype == torch.bfloat16)
            segs[i].pool = int(ss.pool)
            if ss.has_proj:
                w, b = proj[2 * pi], proj[2 * pi + 1]
                if tuple(w.shape) != (d, ss.d_in):
                    raise _lib.EgxError(f"projection {i} weight shape {tuple(w.shape)} != ({d}, {ss.d_in})")
                segs[i].proj_w, segs[i].proj_b = ptr(w), ptr(b)
                pi += 1
            if ss.add_row is not None:
                segs[i].add_vec = _elem_ptr(task_embed, ss.add_row, d)
            if ss.pos_row0 is not None:
                segs[i].pos = _elem_ptr(pos_table, ss.pos_row0, d)
                segs[i].pos_stride = d
        layers = (Layer * max(spec.n_layers, 1))()
        for l in range(spec.n_layers):
            for k, name in enumerate(_LAYER_FIELDS):
                setattr(layers[l], name, ptr(layer_t[12 * l + k]))

        # NB: grad mode is off inside Function.forward; ctx.needs_input_grad is the reliable signal.
        # needs_input_grad order: (spec, task_embed, pos_table, ln_w, ln_b, *rest)
        nig = ctx.needs_input_grad
        needs_grad = any(nig)
        if spec.impl == "auto" and (nig[2] or any(nig[5:5 + nseg])):
            import dataclasses
            feat_grad = [bool(nig[5 + i]) for i in range(nseg)]
            probe_cfg = spec.config()
            wide_ident_only = (lib.egx_encoder_impl(C.byref(probe_cfg), segs, B) == EGX_IMPL_WIDE
                               and not any(fg and ss.has_proj for fg, ss in zip(feat_grad, spec.segments)))
            if any(feat_grad) and not wide_ident_only:
                # gradients into projected features: only the shape-generic backward produces them (the wide path serves
                # identity segments — the trainable action stream of the LTA translators — from its LayerNorm backward)
                spec = dataclasses.replace(spec, impl="generic")
            elif any(feat_grad):
                pass
            # (a learned positional table - the HOI translators' `pe` - gets its gradient from every implementation)
        # first-tokens-only output: in-kernel on the fused path; elsewhere the full block is sliced here (and the gradient
        # scattered back in backward)
        py_slice = 0
        if spec.out_tokens and nhead:
            raise _lib.EgxError("out_tokens (first-tokens-only output) cannot be combined with the fused pooled head: the head pools "
                                "every token of the clip")
        if spec.out_tokens:
            import dataclasses
            full = dataclasses.replace(spec, out_tokens=0)
            probe = full.config()
            if nhead or lib.egx_encoder_impl(C.byref(probe), segs, B) != EGX_IMPL_FUSED:
                py_slice, spec = spec.out_tokens, full
        cfg = spec.config()
        sv, sc = C.c_size_t(0), C.c_size_t(0)
        ws = lib.egx_translator_workspace if nhead else lib.egx_encoder_workspace
        check(ws(C.byref(cfg), segs, B, C.byref(sv), C.byref(sc)))
        S = sum(s.T for s in spec.segments)
        if needs_grad:
            saved = torch.empty(max(sv.value, 256), dtype=torch.uint8, device=device)
            if _POISON:
                saved.fill_(255)
        else:
            saved = _workspace("saved", device, sv.value)
        scratch = _workspace("scratch", device, sc.value)
        seed = C.c_uint64(spec.seed & (2**64 - 1))
        # persistent packed-weight cache: valid when nothing that goes into the packed copies changed since the forward that filled it
        wc, wc_sig = spec.wcache, None
        if wc is not None:
            nbytes = lib.egx_weight_cache_bytes(C.byref(cfg), segs)
            if nbytes:
                packed = proj[0::2] + [layer_t[12 * l + k] for l in range(spec.n_layers) for k in (0, 2, 4, 6)]
                keep = (1.0 / (1.0 - spec.p_drop)) if (spec.training and 0.0 < spec.p_drop < 1.0) else 1.0
                wc_sig = (spec.compute, keep, _weights_epoch[0], nbytes) + tuple((t.data_ptr(), t._version) for t in packed)
                if wc.buf is None or wc.buf.numel() < nbytes or wc.buf.device != device:
                    wc.buf, wc.sig = torch.zeros(nbytes, dtype=torch.uint8, device=device), None
                capturing = torch.cuda.is_current_stream_capturing()
                valid = wc.sig == wc_sig and (wc.frozen or not capturing)
                cfg.weight_cache, cfg.weight_cache_valid = wc.buf.data_ptr(), int(valid)
                wc.hits += int(valid)
                wc.packs += int(not valid)
                wc.sig = None       # (set again once the call has succeeded)
            else:
                wc = None
        ce_keep = None
        if spec.ce:
            if ce_target.dtype != torch.int64 or tuple(ce_target.shape) != (B,) or ce_target.device != device:
                raise ValueError("target must be an int64 tensor of shape (B,) on the features' device")
            tgt = ce_target.contiguous()
            cw = None if ce_weight is None else _dev_f32(ce_weight, "class weight")
            if cw is not None and cw.numel() != spec.head_n_out:
                raise ValueError("class weight must have one entry per class")
            loss = torch.empty((), dtype=torch.float32, device=device)
            dl = torch.empty((B, spec.head_n_out), dtype=torch.float32, device=device)
            ce_struct = Ce(ptr(tgt), ptr(cw), ptr(loss), ptr(dl))
            cfg.ce = C.cast(C.pointer(ce_struct), C.c_void_p)
            ce_keep = (tgt, cw, ce_struct)
        tce_keep = tce_out = None
        if spec.token_ce:
            Cn, Mrows = int(spec.token_ce), B * (spec.out_tokens or S)
            if wc is None or py_slice:
                raise _lib.EgxError("EncoderSpec.token_ce: this configuration does not evaluate the token classifier in its kernels "
                                    "(use functional.encoder_token_ce, which falls back to linear_cross_entropy)")
            tw = _dev_f32(tce_in[0], "classifier weight")
            tb = _dev_f32(tce_in[1], "classifier bias") if tce_in[1] is not None else None
            ttgt, tcw = tce_in[2], (None if tce_in[3] is None else _dev_f32(tce_in[3], "class weight"))
            if tuple(tw.shape) != (Cn, d) or ttgt.dtype != torch.int64 or tuple(ttgt.shape) != (Mrows,) or ttgt.device != device:
                raise ValueError(f"token classifier: weight must be ({Cn}, {d}) and target an int64 tensor of shape ({Mrows},) on the features' device")
            ttgt = ttgt.contiguous()
            t_logits = torch.empty((Mrows, Cn), dtype=torch.float32, device=device)
            t_probs, t_dl = torch.empty_like(t_logits), torch.empty_like(t_logits)
            t_pred = torch.empty((Mrows,), dtype=torch.float32, device=device)
            t_loss = torch.empty((), dtype=torch.float32, device=device)
            t_correct = torch.empty((), dtype=torch.float32, device=device)
            tce_struct = _lib.TokenCe(ptr(tw), ptr(tb), ptr(ttgt), ptr(tcw), Cn, ptr(t_logits), ptr(t_probs), ptr(t_pred), ptr(t_loss),
                                      ptr(t_correct), ptr(t_dl), None, None)
            cfg.token_ce = C.cast(C.pointer(tce_struct), C.c_void_p)
            if not lib.egx_encoder_token_ce_ok(C.byref(cfg), segs, B):
                raise _lib.EgxError("EncoderSpec.token_ce: this configuration does not evaluate the token classifier in its kernels "
                                    "(use functional.encoder_token_ce, which falls back to linear_cross_entropy)")
            tce_keep = (tw, tb, ttgt, tcw, tce_struct)
            tce_out = (t_loss, t_logits, t_probs, t_pred, t_correct)
        if nhead:
            head = Head(ptr(head_t[0]), ptr(head_t[1]), ptr(head_t[2]), ptr(head_t[3]), spec.head_n_out)
            tokens = torch.empty((B, spec.head_n_out), dtype=torch.float32, device=device)   # logits
            check(lib.egx_translator_fwd(C.byref(cfg), segs, ptr(ln_w), ptr(ln_b), layers, C.byref(head), B, ptr(tokens),
                                         None, ptr(saved), ptr(scratch), int(spec.training), seed, _stream()))
            del ce_keep
        else:
            tokens = torch.empty((B, spec.out_tokens or S, d), dtype=torch.float32, device=device)
            check(lib.egx_encoder_fwd(C.byref(cfg), segs, ptr(ln_w), ptr(ln_b), layers, B, ptr(tokens), ptr(saved),
                                      ptr(scratch), int(spec.training), seed, _stream()))
            if py_slice:
                tokens = tokens[:, :py_slice].contiguous()
            del tce_keep
        if wc is not None:
            wc.sig = wc_sig
        ctx.wcache_buf = wc.buf if wc is not None else None     # the backward reads the packed copies from the same buffer
        ctx.py_slice, ctx.S = py_slice, S
        ctx.spec = spec
        ctx.impl = lib.egx_encoder_impl(C.byref(cfg), segs, B)       # EGX_IMPL_FUSED / EGX_IMPL_TILED / EGX_IMPL_WIDE / EGX_IMPL_GENERIC
        _last_impl[0] = ctx.impl
        _last_slices[0] = lib.egx_encoder_slices(C.byref(cfg), segs, B) if ctx.impl == EGX_IMPL_FUSED else 1
        ctx.fused_path = ctx.impl == EGX_IMPL_FUSED
        ctx.B = B
        ctx.nseg, ctx.nproj, ctx.nhead = nseg, nproj, nhead
        ctx.saved_buf = saved
        ctx.scratch_bytes = sc.value
        ctx.has_te = task_embed is not None
        ctx.has_pos = pos_table is not None
        ctx.save_for_backward(*([t for t in (task_embed, pos_table) if t is not None] + [ln_w, ln_b] + feats + proj + layer_t + head_t
                                + ([tw] + ([tb] if tb is not None else []) if spec.token_ce else [])))
        if spec.token_ce:
            ctx.tce_dl, ctx.tce_has_b = t_dl, tb is not None
            ctx.mark_non_differentiable(*tce_out[1:])
            ctx.set_materialize_grads(False)
            return (tokens,) + tce_out
        if spec.ce:
            ctx.ce_dl = dl
            ctx.set_materialize_grads(False)
            return tokens, loss
        return tokens

    @staticmethod
    def backward(ctx, d_tokens, d_loss=None, *_aux):
        spec0: EncoderSpec = ctx.spec
        dl_scale = None
        if spec0.token_ce:
            # (tokens, loss, ...) outputs: the loss's d_logits was left by the forward; the backward's first launch rebuilds d tokens from it
            if d_tokens is not None:
                raise _lib.EgxError("EncoderSpec.token_ce: a gradient through the returned tokens as well as through the fused loss is not supported "
                                    "(compose encoder() and linear_cross_entropy() instead)")
            if d_loss is None:
                return (None,) * len(ctx.needs_input_grad)
            dl_scale = d_loss if (d_loss.dtype == torch.float32 and d_loss.is_contiguous()) else d_loss.float().contiguous()
            d_tokens = ctx.tce_dl       # (only its device is read below)
        if spec0.ce:
            # (logits, loss) outputs: the loss's d_logits was left by the forward; its upstream gradient (loss.backward()'s ones, a loss
            # scale) goes to the kernels as a device scalar. A gradient reaching the logits directly as well is the rare case: torch ops.
            if d_loss is None and d_tokens is None:
                return (None,) * len(ctx.needs_input_grad)
            if d_loss is not None:
                g = d_loss if (d_loss.dtype == torch.float32 and d_loss.is_contiguous()) else d_loss.float().contiguous()
                if d_tokens is None:
                    d_tokens, dl_scale = ctx.ce_dl, g
                else:
                    d_tokens = d_tokens + ctx.ce_dl * g
        if ctx.py_slice:        # gradient of the python-side slice: zeros behind the first tokens
            full = torch.zeros((d_tokens.shape[0], ctx.S, d_tokens.shape[2]), dtype=torch.float32, device=d_tokens.device)
            full[:, :ctx.py_slice] = d_tokens
            d_tokens = full
        lib = _lib.load()
        if reload_tuning_each_call:
            lib.egx_tuning_reload()
        spec: EncoderSpec = ctx.spec
        sv = list(ctx.saved_tensors)
        tce_b = sv.pop() if (spec.token_ce and ctx.tce_has_b) else None
        tce_w = sv.pop() if spec.token_ce else None
        task_embed = sv.pop(0) if ctx.has_te else None
        pos_table = sv.pop(0) if ctx.has_pos else None
        ln_w, ln_b = sv[0], sv[1]
        nseg, nproj = ctx.nseg, ctx.nproj
        feats = sv[2:2 + nseg]
        proj = sv[2 + nseg:2 + nseg + 2 * nproj]
        nhead = ctx.nhead
        layer_t = sv[2 + nseg + 2 * nproj:len(sv) - nhead]
        head_t = sv[len(sv) - nhead:] if nhead else []
        d = spec.d_model
        B = ctx.B
        device = d_tokens.device
        need = ctx.needs_input_grad  # (spec, task_embed, pos_table, ln_w, ln_b, *rest)

        pk = _GradPacker()
        _rr = spec.n_layers if bucket_hook is not None else 0
        i_te = pk.add(task_embed, need[1], rank=_rr)
        i_pos = pk.add(pos_table, need[2], rank=_rr)
        i_lnw = pk.add(ln_w, need[3], rank=_rr)
        i_lnb = pk.add(ln_b, need[4], rank=_rr)
        # d(feature) is a per-sample ACTIVATION gradient: it must never sit in the flat buffer that the data-parallel exchange
        # all-reduces (in place, on the collective's stream, while autograd hands the same memory to the upstream module).
        # Every implementation overwrites it (input-gradient GEMM / LayerNorm backward), so it needs no zero fill either.
        feat_grads = [torch.empty(f.shape, dtype=torch.float32, device=device) if need[5 + i] else None for i, f in enumerate(feats)]
        # proj: (w, b) pairs; layer: 12 tensors in _LAYER_FIELDS order, in_proj_w / out_proj_w are indices 0 and 2
        i_proj = [pk.add(t, need[5 + nseg + i], late=(i % 2 == 0) and bucket_hook is None, rank=_rr) for i, t in enumerate(proj)]
        hook = bucket_hook
        n_l = spec.n_layers
        # bucketed exchange: the backward finishes the LAST layer first -> it gets rank 0, ..., layer 0 rank L - 1, the rest L
        lrank = (lambda i: n_l - 1 - i // 12) if hook is not None else (lambda i: 0)
        rest_rank = n_l if hook is not None else 0
        i_layer = [pk.add(t, need[5 + nseg + 2 * nproj + i], late=(i % 12 in (0, 2)) and hook is None, rank=lrank(i)) for i, t in enumerate(layer_t)]
        i_head = [pk.add(t, need[5 + nseg + 2 * nproj + len(layer_t) + i], rank=rest_rank) for i, t in enumerate(head_t)]
        # the token classifier's gradients (egx_token_ce) live in the flat buffer too: zero-filled, reduced and all-reduced with the rest
        i_tw = pk.add(tce_w, bool(spec.token_ce) and need[-4], rank=rest_rank)
        i_tb = pk.add(tce_b, bool(spec.token_ce) and need[-3], rank=rest_rank)
        grads = pk.materialise(device, zero=False)      # zero-filled by the library's backward (saves a fill launch)

        def g(i):
            return grads[i] if i >= 0 else None

        segs = (Segment * nseg)()
        sgr = (SegmentGrads * nseg)()
        pi = 0
        for i, (ss, f) in enumerate(zip(spec.segments, feats)):
            segs[i].feat = ptr(f)
            segs[i].T = ss.T
            segs[i].d_in = ss.d_in
            segs[i].feat_bf16 = int(f.dtype == torch.bfloat16)
            segs[i].pool = int(ss.pool)
            sgr[i].feat = ptr(feat_grads[i])
            if ss.has_proj:
                segs[i].proj_w, segs[i].proj_b = ptr(proj[2 * pi]), ptr(proj[2 * pi + 1])
                sgr[i].proj_w, sgr[i].proj_b = ptr(g(i_proj[2 * pi])), ptr(g(i_proj[2 * pi + 1]))
                pi += 1
            if ss.add_row is not None:
                segs[i].add_vec = _elem_ptr(task_embed, ss.add_row, d)
                if i_te >= 0:
                    sgr[i].add_vec = _elem_ptr(grads[i_te], ss.add_row, d)
            if ss.pos_row0 is not None:
                segs[i].pos = _elem_ptr(pos_table, ss.pos_row0, d)
                segs[i].pos_stride = d
                if i_pos >= 0:
                    sgr[i].pos = _elem_ptr(grads[i_pos], ss.pos_row0, d)
        layers = (Layer * max(spec.n_layers, 1))()
        lgr = (LayerGrads * max(spec.n_layers, 1))()
        for l in range(spec.n_layers):
            for k, name in enumerate(_LAYER_FIELDS):
                setattr(layers[l], name, ptr(layer_t[12 * l + k]))
                setattr(lgr[l], name, ptr(g(i_layer[12 * l + k])))

        cfg = spec.config()
        cfg.zero_buf, cfg.zero_bytes = ptr(pk.flat), pk.flat.numel() * 4
        if ctx.wcache_buf is not None:
            cfg.weight_cache = ctx.wcache_buf.data_ptr()
        if dl_scale is not None:
            cfg.d_logits_scale = dl_scale.data_ptr()
        tce_keep = tce_grads = None
        if spec.token_ce:
            tce_struct = _lib.TokenCe(ptr(tce_w), None, None, None, int(spec.token_ce), None, None, None, None, None, ptr(ctx.tce_dl),
                                      ptr(g(i_tw)), ptr(g(i_tb)))
            cfg.token_ce = C.cast(C.pointer(tce_struct), C.c_void_p)
            tce_keep, tce_grads = tce_struct, (g(i_tw), g(i_tb))
        announced = set()
        cb_keep = None
        if hook is not None and ctx.impl == EGX_IMPL_WIDE:
            flat_for_cb, spans = pk.flat, dict(pk.rank_span)

            cb_error = []

            def _cb(_user, bucket):       # host callback from wide_encoder_bwd: layer n_l - 1 - bucket is complete on the stream
                # ctypes swallows an exception raised in a callback ("Exception ignored"): keep it, leave the bucket
                # un-announced (the remainder pass below then covers its slice) and re-raise once the launch has returned
                if cb_error or bucket not in spans or bucket in announced:
                    return
                try:
                    hook(flat_for_cb, spans[bucket][0], spans[bucket][1])
                    announced.add(bucket)
                except BaseException as e:     # noqa: BLE001
                    cb_error.append(e)
            cb_keep = _lib.BUCKET_CB(_cb)
            cfg.bucket_cb = C.cast(cb_keep, C.c_void_p)
        defer = bool(spec.defer_small) and ctx.fused_path
        cfg.bwd_stage = 1 if defer else 0
        scratch = _workspace("scratch", device, ctx.scratch_bytes)
        seed = C.c_uint64(spec.seed & (2**64 - 1))
        saved_buf = ctx.saved_buf
        # raw addresses only inside `launch`: a closure that kept the gradient VIEW tensors alive would make autograd
        # clone them instead of adopting them as .grad (they must stay views of the flat buffer)
        p_lnw, p_lnb, flat_buf = ptr(g(i_lnw)), ptr(g(i_lnb)), pk.flat
        if nhead:
            head = Head(ptr(head_t[0]), ptr(head_t[1]), ptr(head_t[2]), ptr(head_t[3]), spec.head_n_out)
            hg = HeadGrads(ptr(g(i_head[0])), ptr(g(i_head[1])), ptr(g(i_head[2])), ptr(g(i_head[3])))
            dtok = d_tokens.float().contiguous()

            def launch(c):
                check(lib.egx_translator_bwd(C.byref(c), segs, ptr(ln_w), ptr(ln_b), layers, C.byref(head), B, ptr(dtok),
                                             ptr(saved_buf), ptr(scratch), sgr, p_lnw, p_lnb, lgr,
                                             C.byref(hg), int(spec.training), seed, _stream()))
        else:
            # the generic backward overwrites d_tokens (needs a private copy); the fused and wide kernels only read it
            if spec.token_ce:
                dtok = None
            else:
                dtok = d_tokens if d_tokens.dtype == torch.float32 else d_tokens.float()
                dtok = dtok.contiguous()
                if dtok.data_ptr() == d_tokens.data_ptr() and ctx.impl == EGX_IMPL_GENERIC:
                    dtok = dtok.clone()

            def launch(c):
                check(lib.egx_encoder_bwd(C.byref(c), segs, ptr(ln_w), ptr(ln_b), layers, B, ptr(dtok), ptr(saved_buf),
                                          ptr(scratch), sgr, p_lnw, p_lnb, lgr, int(spec.training), seed,
                                          _stream()))
        launch(cfg)
        if cb_keep is not None and cb_error:
            raise cb_error[0]
        if hook is not None and not defer:
            # whatever the per-layer callbacks did not announce (other implementations: everything) is complete now
            done = sorted(pk.rank_span[b] for b in announced)
            cur = 0
            for lo, hi in done + [(pk.total, pk.total)]:
                if lo > cur:
                    hook(pk.flat, cur, lo)
                cur = max(cur, hi)
        last_grad_layout.update(flat=pk.flat, late_floats=pk.late_floats if defer else 0)
        if defer:
            # stage 2 (dW_proj, dW_in, dW_o) on request: same arguments, kept alive by this closure; the shared scratch
            # workspace must not be reused by another encoder call before run_deferred()
            cfg2 = spec.config()
            cfg2.bwd_stage = 2
            if ctx.wcache_buf is not None:
                cfg2.weight_cache = ctx.wcache_buf.data_ptr()
            keep = (flat_buf, sv, dtok, scratch, saved_buf)   # noqa: F841  (referenced by `finish`: keeps the buffers alive)

            def finish(_keep=keep):
                launch(cfg2)
            _deferred.append(finish)
        out = [None, g(i_te), g(i_pos), g(i_lnw), g(i_lnb)]
        out += feat_grads + [g(i) for i in i_proj] + [g(i) for i in i_layer] + [g(i) for i in i_head]
        if spec.ce:
            out += [None, None]     # target, class weight
        if spec.token_ce:
            nig = ctx.needs_input_grad
            out += [tce_grads[0], tce_grads[1], None, None]
            del tce_keep
        return tuple(out)


def encoder(spec: EncoderSpec, feats: Sequence[torch.Tensor], task_embed, pos_table, ln_w, ln_b,
            proj: Sequence[torch.Tensor], layer_params: Sequence[torch.Tensor],
            head_params: Sequence[torch.Tensor] = (), ce=None):
    """head_params = (head_ln_w, head_ln_b, head_W, head_b) with spec.head_n_out = head_W.shape[0] -> logits.
    ce = (target (B,) int64, class_weight | None) with spec.ce: -> (logits, loss), the weighted cross entropy evaluated by the forward."""
    if spec.ce:
        return EncoderFn.apply(spec, task_embed, pos_table, ln_w, ln_b, *feats, *proj, *layer_params, *head_params, ce[0], ce[1])
    return EncoderFn.apply(spec, task_embed, pos_table, ln_w, ln_b, *feats, *proj, *layer_params, *head_params)


def _token_ce_fused(spec: EncoderSpec, feats, proj) -> bool:
    """Does this configuration evaluate EncoderSpec.token_ce in its kernels (egx_encoder_token_ce_ok)? Probed with the real feature / projection
    tensors; the weight cache only has to exist."""
    if spec.wcache is None or spec.deterministic or not spec.out_tokens:
        return False
    lib = _lib.load()
    nseg = len(spec.segments)
    segs = (Segment * nseg)()
    pi = 0
    for i, (ss, f) in enumerate(zip(spec.segments, feats)):
        if not (f.is_cuda and f.dim() == 3):
            return False
        segs[i].feat, segs[i].T, segs[i].d_in = ptr(f), ss.T, ss.d_in
        segs[i].feat_bf16, segs[i].pool = int(f.dtype == torch.bfloat16), int(ss.pool)
        if ss.has_proj:
            segs[i].proj_w, segs[i].proj_b = ptr(proj[2 * pi]), ptr(proj[2 * pi + 1])
            pi += 1
    cfg = spec.config()
    if lib.egx_encoder_impl(C.byref(cfg), segs, feats[0].shape[0]) != EGX_IMPL_FUSED:
        return False
    cfg.weight_cache = 1        # (any non-null value: the query does not touch it)
    return bool(lib.egx_encoder_token_ce_ok(C.byref(cfg), segs, feats[0].shape[0]))


def encoder_token_ce(spec: EncoderSpec, feats: Sequence[torch.Tensor], task_embed, pos_table, ln_w, ln_b,
                     proj: Sequence[torch.Tensor], layer_params: Sequence[torch.Tensor], fc_w, fc_b, target, class_weight=None):
    """(loss, logits, probs, pred, correct) of Linear(d -> C) + weighted cross entropy on the first spec.out_tokens tokens of every clip (rows in
    clip-major order, as encoder(...).reshape(B * out_tokens, d) has them): the ASD task's lossAV on the translator's per-frame output
    (HHI/tasks/asd/video_task_taskspecific.py:24,33). Where the per-clip kernels can, the encoder launches evaluate it themselves
    (egx_token_ce: two launches less per step); elsewhere encoder() + linear_cross_entropy()."""
    import dataclasses
    if _token_ce_fused(spec, feats, proj):
        spec = dataclasses.replace(spec, token_ce=int(fc_w.shape[0]))
        out = EncoderFn.apply(spec, task_embed, pos_table, ln_w, ln_b, *feats, *proj, *layer_params, fc_w, fc_b, target, class_weight)
        return out[1:]
    tokens = encoder(spec, feats, task_embed, pos_table, ln_w, ln_b, proj, layer_params)
    return linear_cross_entropy(tokens.reshape(-1, tokens.shape[-1]), fc_w, fc_b, target, class_weight)


class PoolHeadFn(torch.autograd.Function):
    """out = [Linear]([LN](mean_s tokens)); ln / linear are optional (None)."""

    @staticmethod
    def forward(ctx, tokens, ln_w, ln_b, W, b, eps: float):
        lib = _lib.load()
        tokens = _dev_f32(tokens, "tokens")
        B, S, d = tokens.shape
        ln_w = _dev_f32(ln_w, "ln weight") if ln_w is not None else None
        ln_b = _dev_f32(ln_b, "ln bias") if ln_b is not None else None
        W = _dev_f32(W, "head weight") if W is not None else None
        b = _dev_f32(b, "head bias") if b is not None else None
        n_out = W.shape[0] if W is not None else d
        pooled = torch.empty((B, d), dtype=torch.float32, device=tokens.device)
        out = torch.empty((B, n_out), dtype=torch.float32, device=tokens.device)
        check(lib.egx_pool_head_fwd(ptr(tokens), B, S, d, ptr(ln_w), ptr(ln_b), float(eps), ptr(W), ptr(b), n_out,
                                    ptr(pooled), ptr(out), _stream()))
        ctx.dims = (B, S, d, n_out, float(eps))
        ctx.has = (ln_w is not None, W is not None, b is not None)
        ctx.save_for_backward(*[t for t in (pooled, ln_w, ln_b, W) if t is not None])
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        B, S, d, n_out, eps = ctx.dims
        has_ln, has_W, has_b = ctx.has
        sv = list(ctx.saved_tensors)
        pooled = sv.pop(0)
        ln_w = sv.pop(0) if has_ln else None
        ln_b = sv.pop(0) if has_ln else None
        W = sv.pop(0) if has_W else None
        d_out = d_out.contiguous().float()
        dev = d_out.device
        d_tokens = torch.empty((B, S, d), dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad
        pk = _GradPacker()
        i_lw = pk.add(ln_w, has_ln and need[1])
        i_lb = pk.add(ln_b, has_ln and need[2])
        i_W = pk.add(W, has_W and need[3])
        i_b = pk.add(torch.empty(n_out, device="meta") if has_b else None, has_b and need[4])
        gr = pk.materialise(dev)

        def g(i):
            return gr[i] if i >= 0 else None

        check(lib.egx_pool_head_bwd(ptr(d_out), ptr(pooled), B, S, d, ptr(ln_w), ptr(ln_b), eps, ptr(W), n_out,
                                    ptr(d_tokens), ptr(g(i_lw)), ptr(g(i_lb)), ptr(g(i_W)), ptr(g(i_b)), _stream()))
        return (d_tokens if need[0] else None), g(i_lw), g(i_lb), g(i_W), g(i_b), None


def pool_head(tokens, ln_w=None, ln_b=None, W=None, b=None, eps: float = 1e-5):
    return PoolHeadFn.apply(tokens, ln_w, ln_b, W, b, eps)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for a 2-D x, through the MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, W, b, compute: str, relu: bool = False):
        lib = _lib.load()
        x = _dev_f32(x, "x")
        W = _dev_f32(W, "W")
        b = _dev_f32(b, "b") if b is not None else None
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        check(lib.egx_linear_fwd(ptr(x), ptr(W), ptr(b), ptr(y), M, N, K, int(relu), COMPUTE[compute], _stream()))
        ctx.compute = compute
        ctx.has_b = b is not None
        ctx.relu = bool(relu)
        ctx.save_for_backward(*((x, W, y) if relu else (x, W)))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, W = ctx.saved_tensors[:2]
        dy = dy.contiguous().float()
        if ctx.relu:                                  # d relu: zero the gradient where the output was clamped
            dy = dy.clone()
            check(lib.egx_relu_mask(ptr(dy), ptr(ctx.saved_tensors[2]), dy.numel(), _stream()))
        M, K = x.shape
        N = W.shape[0]
        need = ctx.needs_input_grad
        dev = dy.device
        dx = torch.empty_like(x) if need[0] else None
        dW = torch.zeros_like(W) if need[1] else None
        db = torch.zeros(N, dtype=torch.float32, device=dev) if (ctx.has_b and need[2]) else None
        nbytes = lib.egx_linear_bwd_scratch(M, N, K)
        scratch = _workspace("linear", dev, nbytes)
        check(lib.egx_linear_bwd(ptr(dy), ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), M, N, K, COMPUTE[ctx.compute],
                                 ptr(scratch), _stream()))
        return dx, dW, db, None, None


class StackedLinearFn(torch.autograd.Function):
    """y = x Wst^T + bst where Wst / bst are the row-wise stack of several nn.Linear parameters that LIVE in that stack (their `.data` are views
    of it: hoi_lta.MultiTaskHead). One GEMM forward, one backward; the weight / bias gradients come back as views of ONE stacked gradient
    buffer each — no torch.cat of twenty weight matrices per step, no split of the gradient.
    Arguments: x (M, K), Wst (N, K), bst (N), sizes, compute, then the member weights and the member biases (gradient routing only)."""

    @staticmethod
    def forward(ctx, x, Wst, bst, sizes, compute: str, *members):
        lib = _lib.load()
        x = _dev_f32(x, "x")
        M, K = x.shape
        N = Wst.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        check(lib.egx_linear_fwd(ptr(x), ptr(Wst), ptr(bst), ptr(y), M, N, K, 0, COMPUTE[compute], _stream()))
        ctx.compute, ctx.sizes, ctx.nm = compute, tuple(sizes), len(members) // 2
        ctx.save_for_backward(x, Wst)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, Wst = ctx.saved_tensors
        dy = dy.contiguous().float()
        M, K = x.shape
        N = Wst.shape[0]
        need = ctx.needs_input_grad
        dev = dy.device
        dx = torch.empty_like(x) if need[0] else None
        want_w = any(need[5:5 + ctx.nm])
        want_b = any(need[5 + ctx.nm:])
        dW = torch.zeros_like(Wst) if want_w else None
        db = torch.zeros(N, dtype=torch.float32, device=dev) if want_b else None
        scratch = _workspace("linear", dev, lib.egx_linear_bwd_scratch(M, N, K))
        check(lib.egx_linear_bwd(ptr(dy), ptr(x), ptr(Wst), ptr(dx), ptr(dW), ptr(db), M, N, K, COMPUTE[ctx.compute],
                                 ptr(scratch), _stream()))
        gw, gb, off = [], [], 0
        for i, n in enumerate(ctx.sizes):
            gw.append(dW[off:off + n] if (dW is not None and need[5 + i]) else None)
            gb.append(db[off:off + n] if (db is not None and need[5 + ctx.nm + i]) else None)
            off += n
        return (dx, None, None, None, None, *gw, *gb)


def linear(x, W, b=None, compute: str = "f32", relu: bool = False):
    shp = x.shape
    y = LinearFn.apply(x.reshape(-1, shp[-1]), W, b, compute, relu)
    return y.view(*shp[:-1], W.shape[0])


class LinearResidualFn(torch.autograd.Function):
    """y = x W^T (+ b) + residual: the residual connection rides in the GEMM epilogue (pre-LN blocks)."""

    @staticmethod
    def forward(ctx, x, W, b, residual, compute: str):
        lib = _lib.load()
        x, W, residual = _dev_f32(x, "x"), _dev_f32(W, "W"), _dev_f32(residual, "residual")
        b = _dev_f32(b, "b") if b is not None else None
        M, K = x.shape
        N = W.shape[0]
        if tuple(residual.shape) != (M, N):
            raise ValueError(f"residual shape {tuple(residual.shape)} != ({M}, {N})")
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        check(lib.egx_linear_residual_fwd(ptr(x), ptr(W), ptr(b), ptr(residual), ptr(y), M, N, K, COMPUTE[compute], _stream()))
        ctx.compute, ctx.has_b = compute, b is not None
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        dy = dy.contiguous().float()
        M, K = x.shape
        N = W.shape[0]
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dW = torch.zeros_like(W) if need[1] else None
        db = torch.zeros(N, dtype=torch.float32, device=dy.device) if (ctx.has_b and need[2]) else None
        scratch = _workspace("linear", dy.device, lib.egx_linear_bwd_scratch(M, N, K))
        check(lib.egx_linear_bwd(ptr(dy), ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), M, N, K, COMPUTE[ctx.compute],
                                 ptr(scratch), _stream()))
        return dx, dW, db, (dy if need[3] else None), None


def linear_residual(x, W, b, residual, compute: str = "f32"):
    return LinearResidualFn.apply(x, W, b, residual, compute)


class GeluFn(torch.autograd.Function):
    """Exact (erf) GELU; the pre-activation is kept for the backward."""

    @staticmethod
    def forward(ctx, z):
        lib = _lib.load()
        z = _dev_f32(z, "z")
        h = torch.empty_like(z)
        check(lib.egx_gelu_fwd(ptr(z), ptr(h), z.numel(), _stream()))
        ctx.save_for_backward(z)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        (z,) = ctx.saved_tensors
        dh = dh.contiguous().float()
        dz = torch.empty_like(z)
        check(lib.egx_gelu_bwd(ptr(z), ptr(dh), ptr(dz), z.numel(), _stream()))
        return dz


def gelu(z):
    return GeluFn.apply(z)


class AttentionFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(dh)) V per (clip, head) from packed (B * S, 3 * inner) qkv rows -> (B * S, inner), through the
    shape-generic attention kernels (any S; inner = H * dh need not equal the model width)."""

    @staticmethod
    def forward(ctx, qkv, B: int, S: int, H: int):
        lib = _lib.load()
        qkv = _dev_f32(qkv, "qkv")
        inner = qkv.shape[1] // 3
        out = torch.empty((B * S, inner), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
        check(lib.egx_attention_fwd(ptr(qkv), ptr(out), ptr(lse), B, S, H, inner, 0.0, C.c_uint64(0), _stream()))
        ctx.dims = (B, S, H, inner)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        qkv, out, lse = ctx.saved_tensors
        B, S, H, inner = ctx.dims
        d_out = d_out.contiguous().float()
        dqkv = torch.empty_like(qkv)
        check(lib.egx_attention_bwd(ptr(qkv), ptr(out), ptr(lse), ptr(d_out), ptr(dqkv), B, S, H, inner, 0.0, C.c_uint64(0), _stream()))
        return dqkv, None, None, None


def attention(qkv, B: int, S: int, H: int):
    return AttentionFn.apply(qkv, B, S, H)


_UNIT_GRAD = {}


def unit_grad(device) -> torch.Tensor:
    """A persistent scalar 1.0 on `device` for `loss.backward(gradient=unit_grad(dev))`: autograd then needs no
    ones-fill, and weighted_cross_entropy recognises it and skips the multiplication by 1 (two launches per step)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    t = _UNIT_GRAD.get(key)
    if t is None:
        t = _UNIT_GRAD[key] = torch.ones((), dtype=torch.float32, device=torch.device("cuda", key))
    return t


class WeightedCEFn(torch.autograd.Function):
    """loss = sum_i w[y_i] nll_i / sum_i w[y_i]; the same launch leaves d loss / d logits behind for backward."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        lib = _lib.load()
        z = _dev_f32(logits, "logits")
        if z.dim() != 2:
            raise ValueError(f"weighted_cross_entropy expects (B, C) logits, got {tuple(z.shape)}")
        if target.dtype != torch.int64 or target.shape != z.shape[:1] or target.device != z.device:
            raise ValueError("target must be an int64 tensor of shape (B,) on the logits' device")
        tgt = target.contiguous()
        w = None if weight is None else _dev_f32(weight, "weight")
        if w is not None and w.numel() != z.shape[1]:
            raise ValueError("weight must have one entry per class")
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        dl = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        check(lib.egx_weighted_ce(ptr(z), ptr(tgt), ptr(w) if w is not None else None, z.shape[0], z.shape[1],
                                  ptr(loss), ptr(dl) if dl is not None else None, _stream()))
        ctx.has_dl = dl is not None
        if dl is not None:
            ctx.save_for_backward(dl)       # autograd owns it: a second backward (retain_graph) sees the same gradient,
        return loss                         # a backward after the buffers were freed raises as for any other op

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.has_dl:
            return None, None, None
        (dl,) = ctx.saved_tensors
        u = _UNIT_GRAD.get(dl.device.index)
        if u is not None and grad_out.data_ptr() == u.data_ptr():
            return dl, None, None
        return dl * grad_out, None, None


class DecoderFn(torch.autograd.Function):
    """The EgoT2-g sequence decoder + vocabulary head as ONE library call per direction (egx_decoder_fwd / egx_decoder_bwd, compute =
    bf16): logits (B * sy, |V|) from target tokens (B, sy) and the encoder memory (B * S, d). Inputs after `meta`: memory, embedding
    weight, positional rows (sy, d), 18 tensors per layer in _lib._DEC_LAYER_FIELDS order, fc weight, fc bias. All parameter gradients
    are views of one flat buffer (zero-filled by the library's backward)."""

    @staticmethod
    def forward(ctx, meta, tokens, memory, emb, pe, *rest):
        lib = _lib.load()
        n_layers = meta["n_layers"]
        layer_t = [_dev_f32(t, "decoder layer parameter") for t in rest[:18 * n_layers]]
        fc_w, fc_b = _dev_f32(rest[18 * n_layers], "fc.weight"), _dev_f32(rest[18 * n_layers + 1], "fc.bias")
        memory, emb, pe = _dev_f32(memory, "memory"), _dev_f32(emb, "embedding.weight"), _dev_f32(pe, "positional rows")
        if tokens.dtype != torch.int64 or not tokens.is_cuda:
            raise _lib.EgxError("decoder tokens must be an int64 tensor on the GPU")
        tokens = tokens.contiguous()
        B, sy = tokens.shape
        d = emb.shape[1]
        S = memory.shape[0] // B
        cfg = _lib.DecConfig(d, meta["n_heads"], meta["d_ff"], n_layers, emb.shape[0], sy, S, meta["ln_eps"], EGX_BF16,
                             meta["p_drop"], meta["p_pos"], meta.get("seed_ptr") or None)
        sv, sc = C.c_size_t(0), C.c_size_t(0)
        check(lib.egx_decoder_workspace(C.byref(cfg), B, C.byref(sv), C.byref(sc)))
        need_grad = any(ctx.needs_input_grad)
        saved = torch.empty(max(sv.value, 256), dtype=torch.uint8, device=memory.device) if need_grad else _workspace("dec_saved", memory.device, sv.value)
        scratch = _workspace("dec_scratch", memory.device, sc.value)
        layers = (_lib.DecLayer * n_layers)()
        for l in range(n_layers):
            for k, name in enumerate(_lib._DEC_LAYER_FIELDS):
                setattr(layers[l], name, ptr(layer_t[18 * l + k]))
        logits = torch.empty((B * sy, emb.shape[0]), dtype=torch.float32, device=memory.device)
        seed = C.c_uint64(meta["seed"] & (2**64 - 1))
        check(lib.egx_decoder_fwd(C.byref(cfg), ptr(tokens), ptr(memory), ptr(emb), ptr(pe), pe.stride(0), layers, ptr(fc_w), ptr(fc_b), B,
                                  ptr(logits), ptr(saved), ptr(scratch), int(meta["training"]), seed, _stream()))
        ctx.meta, ctx.cfg_args, ctx.B, ctx.saved_buf, ctx.scratch_bytes = meta, (d, sy, S), B, saved, sc.value
        ctx.save_for_backward(tokens, memory, emb, pe, *layer_t, fc_w, fc_b)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        lib = _lib.load()
        meta = ctx.meta
        n_layers = meta["n_layers"]
        sv = list(ctx.saved_tensors)
        tokens, memory, emb, pe = sv[:4]
        layer_t = sv[4:4 + 18 * n_layers]
        fc_w, fc_b = sv[4 + 18 * n_layers], sv[5 + 18 * n_layers]
        d, sy, S = ctx.cfg_args
        B = ctx.B
        need = ctx.needs_input_grad      # (meta, tokens, memory, emb, pe, *layer_t, fc_w, fc_b)
        device = d_logits.device
        pk = _GradPacker()
        i_emb = pk.add(emb, need[3])
        i_layer = [pk.add(t, need[5 + i]) for i, t in enumerate(layer_t)]
        i_fcw = pk.add(fc_w, need[5 + 18 * n_layers])
        i_fcb = pk.add(fc_b, need[6 + 18 * n_layers])
        grads = pk.materialise(device, zero=False)

        def g(i):
            return grads[i] if i >= 0 else None

        d_memory = torch.empty_like(memory) if need[2] else None
        cfg = _lib.DecConfig(d, meta["n_heads"], meta["d_ff"], n_layers, emb.shape[0], sy, S, meta["ln_eps"], EGX_BF16,
                             meta["p_drop"], meta["p_pos"], meta.get("seed_ptr") or None)
        layers = (_lib.DecLayer * n_layers)()
        lgr = (_lib.DecLayerGrads * n_layers)()
        for l in range(n_layers):
            for k, name in enumerate(_lib._DEC_LAYER_FIELDS):
                setattr(layers[l], name, ptr(layer_t[18 * l + k]))
                setattr(lgr[l], name, ptr(g(i_layer[18 * l + k])))
        scratch = _workspace("dec_scratch", device, ctx.scratch_bytes)
        dl = d_logits.float().contiguous()
        seed = C.c_uint64(meta["seed"] & (2**64 - 1))
        check(lib.egx_decoder_bwd(C.byref(cfg), ptr(tokens), layers, ptr(fc_w), B, ptr(dl), ptr(ctx.saved_buf), ptr(scratch), ptr(d_memory),
                                  ptr(g(i_emb)), lgr, ptr(g(i_fcw)), ptr(g(i_fcb)), ptr(pk.flat), pk.flat.numel() * 4, int(meta["training"]),
                                  seed, _stream()))
        if bucket_hook is not None and pk.total:
            bucket_hook(pk.flat, 0, pk.total)        # the decoder's gradients are exchanged while the encoder's backward runs
        return (None, None, d_memory, g(i_emb), None) + tuple(g(i) for i in i_layer) + (g(i_fcw), g(i_fcb))


def decoder_supported(compute: str, d: int, n_heads: int, d_ff: int, sy: int, S: int, n_layers: int) -> bool:
    """Shapes egx_decoder_fwd / egx_decoder_bwd serve (include/egot2x.h); everything else stays on the composed decoder."""
    return (compute == "bf16" and 256 <= d <= 1024 and d % 128 == 0 and n_heads > 0 and d % n_heads == 0 and d // n_heads in (32, 64)
            and d_ff >= 128 and d_ff % 128 == 0 and 1 <= sy <= 8 and 1 <= S <= 1024 and 1 <= n_layers <= 16)


def weighted_cross_entropy(logits, target, weight=None):
    """F.cross_entropy(logits, target, weight=weight) with mean reduction (HHI/tasks/ttm/video_task_2loader.py:21-22,34).
    Labels outside [0, C) - including F.cross_entropy's default ignore_index = -100 - contribute neither loss, weight nor
    gradient."""
    return WeightedCEFn.apply(logits, target, weight)


_LCE_SCRATCH = {}
_LCE_RETIRED = []       # outgrown buffers stay alive: a hipGraph captured earlier still holds their addresses


def _lce_scratch(device, nbytes: int) -> torch.Tensor:
    """Scratch of the fused classifier head: arrival counters (zero between launches) + partial sums. One per device (NOT per
    stream: a hipGraph is captured on a stream of its own and must find the buffer its eager warm-up created — so two streams
    must not run lossAV concurrently on one device), grown on demand; a new buffer is zero-filled once (the kernels leave the
    counters zero again). An outgrown buffer is never freed: a graph captured at the smaller size keeps replaying into it."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _LCE_SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # a buffer created inside a capture would live in that graph's private pool
            raise RuntimeError("linear_cross_entropy: run one eager step at this problem size before capturing it in a graph")
        if t is not None:
            _LCE_RETIRED.append(t)
        t = _LCE_SCRATCH[key] = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    return t


class LinearCEFn(torch.autograd.Function):
    """(loss, logits, probs, pred, correct) of Linear(K -> C) followed by weighted cross-entropy, ONE launch forward and ONE backward
    (egx_linear_ce_*): the ASD task's lossAV, HHI/tasks/asd/loss.py:11-30. Only `loss` carries gradient."""

    @staticmethod
    def forward(ctx, x, W, b, target, weight):
        lib = _lib.load()
        x, W = _dev_f32(x, "x"), _dev_f32(W, "weight")
        if x.dim() != 2 or W.dim() != 2 or W.shape[1] != x.shape[1]:
            raise ValueError(f"linear_cross_entropy expects x (M, K) and W (C, K), got {tuple(x.shape)} and {tuple(W.shape)}")
        M, K = x.shape
        Cn = W.shape[0]
        if Cn > 8 or K not in (64, 128, 256):
            raise ValueError(f"linear_cross_entropy supports C <= 8 classes and K in (64, 128, 256), got C={Cn}, K={K}")
        if target.dtype != torch.int64 or target.shape != (M,) or target.device != x.device:
            raise ValueError("target must be an int64 tensor of shape (M,) on the input's device")
        b = _dev_f32(b, "bias") if b is not None else None
        w = _dev_f32(weight, "class weight") if weight is not None else None
        if w is not None and w.numel() != Cn:
            raise ValueError("weight must have one entry per class")
        tgt = target.contiguous()
        logits = torch.empty((M, Cn), dtype=torch.float32, device=x.device)
        probs = torch.empty_like(logits)
        need = any(ctx.needs_input_grad[:3])
        dl = torch.empty_like(logits) if need else None
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        correct = torch.empty((), dtype=torch.float32, device=x.device)
        pred = torch.empty((M,), dtype=torch.float32, device=x.device)
        scratch = _lce_scratch(x.device, lib.egx_linear_ce_scratch(M, K, Cn))
        check(lib.egx_linear_ce_fwd(ptr(x), ptr(W), ptr(b) if b is not None else None, ptr(tgt), ptr(w) if w is not None else None,
                                    M, K, Cn, ptr(logits), ptr(probs), ptr(dl) if dl is not None else None, ptr(loss), ptr(correct),
                                    ptr(pred), ptr(scratch), _stream()))
        ctx.has_b = b is not None
        ctx.shape = (M, K, Cn)
        if need:
            ctx.save_for_backward(x, W, dl)
        ctx.mark_non_differentiable(logits, probs, pred, correct)
        ctx.set_materialize_grads(False)    # no zero-filled "gradients" for the four auxiliary outputs (a fill launch each)
        return loss, logits, probs, pred, correct

    @staticmethod
    def backward(ctx, g_loss, _gl, _gp, _gr, _gc):
        if g_loss is None:
            return None, None, None, None, None
        lib = _lib.load()
        x, W, dl = ctx.saved_tensors
        M, K, Cn = ctx.shape
        g = g_loss.contiguous().float()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        db = torch.empty((Cn,), dtype=torch.float32, device=x.device) if ctx.has_b else None
        scratch = _lce_scratch(x.device, lib.egx_linear_ce_scratch(M, K, Cn))
        check(lib.egx_linear_ce_bwd(ptr(x), ptr(W), ptr(dl), ptr(g), M, K, Cn, ptr(dx) if dx is not None else None, ptr(dW),
                                    ptr(db) if db is not None else None, ptr(scratch), _stream()))
        return dx, dW if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None, None, None


def linear_cross_entropy(x, W, b, target, weight=None):
    """(loss, logits, probs, pred, correct): F.cross_entropy(F.linear(x, W, b), target, weight=weight) plus softmax(logits),
    round(softmax)[:, 1] and the number of rows where that equals the label, in one launch (lossAV, HHI/tasks/asd/loss.py:17-30)."""
    return LinearCEFn.apply(x, W, b, target, weight)


# ---- EgoT2-g sequence decoder pieces (kernels in csrc/decoder.hip) ------------------------------
class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) * w + b over the last dimension of 2-D inputs (post-LN residual blocks)."""

    @staticmethod
    def forward(ctx, x, res, w, b, eps: float):
        lib = _lib.load()
        x = _dev_f32(x, "x")
        res = _dev_f32(res, "res") if res is not None else None
        w, b = _dev_f32(w, "ln weight"), _dev_f32(b, "ln bias")
        rows, d = x.shape
        pre = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check(lib.egx_layernorm_fwd(ptr(x), ptr(res) if res is not None else None, ptr(w), ptr(b), float(eps), ptr(pre),
                                    ptr(stats), ptr(y), rows, d, _stream()))
        ctx.has_res = res is not None
        ctx.save_for_backward(pre, stats, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        pre, stats, w = ctx.saved_tensors
        dy = dy.contiguous().float()
        rows, d = pre.shape
        need = ctx.needs_input_grad
        dx = torch.empty_like(pre)
        dw = torch.zeros(d, dtype=torch.float32, device=dy.device) if need[2] else None
        db = torch.zeros(d, dtype=torch.float32, device=dy.device) if need[3] else None
        check(lib.egx_layernorm_bwd(ptr(dy), ptr(pre), ptr(stats), ptr(w), ptr(dx), ptr(dw) if dw is not None else None,
                                    ptr(db) if db is not None else None, rows, d, _stream()))
        return (dx if need[0] else None), (dx if (ctx.has_res and need[1]) else None), dw, db, None


def layer_norm_residual(x, res, w, b, eps: float = 1e-5):
    return LayerNormFn.apply(x, res, w, b, eps)


class DropoutFn(torch.autograd.Function):
    """Inverted dropout with the library's counter-based mask (seed, site, row, column); the backward regenerates it."""

    @staticmethod
    def forward(ctx, x, p: float, seed: int, site: int):
        lib = _lib.load()
        y = _dev_f32(x, "x").clone()
        rows = y.numel() // y.shape[-1]
        check(lib.egx_dropout(ptr(y), rows, y.shape[-1], float(p), C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        ctx.cfg = (float(p), seed, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        p, seed, site = ctx.cfg
        g = dy.contiguous().float().clone()
        rows = g.numel() // g.shape[-1]
        check(lib.egx_dropout(ptr(g), rows, g.shape[-1], p, C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        return g, None, None, None


def dropout(x, p: float, training: bool, seed: int, site: int):
    return DropoutFn.apply(x, p, seed, site) if (training and p > 0.0) else x


class EmbedPosFn(torch.autograd.Function):
    """(B, sy) int64 tokens -> (B * sy, d): embedding[token] * scale + pe[t] (+ dropout)."""

    @staticmethod
    def forward(ctx, tokens, emb, pe, scale: float, p: float, seed: int):
        lib = _lib.load()
        emb = _dev_f32(emb, "embedding")
        pe2 = _dev_f32(pe, "pe")
        if tokens.dtype != torch.int64 or tokens.dim() != 2 or tokens.device != emb.device:
            raise ValueError("tokens must be a (B, sy) int64 tensor on the embedding's device")
        tok = tokens.contiguous()
        B, sy = tok.shape
        V, d = emb.shape
        if sy > pe2.shape[0]:
            raise ValueError("target longer than the positional table")
        out = torch.empty((B * sy, d), dtype=torch.float32, device=emb.device)
        check(lib.egx_embed_pos_fwd(ptr(tok), ptr(emb), ptr(pe2), pe2.stride(0), float(scale), ptr(out), B, sy, d, V, float(p),
                                    C.c_uint64(seed & (2**64 - 1)), _stream()))
        ctx.cfg = (float(scale), float(p), seed, V)
        ctx.save_for_backward(tok)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (tok,) = ctx.saved_tensors
        scale, p, seed, V = ctx.cfg
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None, None
        dy = dy.contiguous().float()
        B, sy = tok.shape
        d = dy.shape[-1]
        d_emb = torch.zeros((V, d), dtype=torch.float32, device=dy.device)
        check(lib.egx_embed_pos_bwd(ptr(tok), ptr(dy), ptr(d_emb), scale, B, sy, d, V, p, C.c_uint64(seed & (2**64 - 1)), _stream()))
        return None, d_emb, None, None, None, None


class SelfAttnSmallFn(torch.autograd.Function):
    """Causal self-attention over a few target tokens from packed (B * sy, 3d) qkv rows -> (B * sy, d)."""

    @staticmethod
    def forward(ctx, qkv, B: int, sy: int, H: int, causal: bool, p: float, seed: int, site: int):
        lib = _lib.load()
        qkv = _dev_f32(qkv, "qkv")
        d = qkv.shape[1] // 3
        out = torch.empty((B * sy, d), dtype=torch.float32, device=qkv.device)
        e = qkv.element_size()
        base = qkv.data_ptr()
        check(lib.egx_small_attention_fwd(base, 3 * d, base + d * e, 3 * d, base + 2 * d * e, 3 * d, ptr(out), d, B, sy, sy, H,
                                          d // H, int(causal), float(p), C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        ctx.cfg = (B, sy, H, int(causal), float(p), seed, site)
        ctx.save_for_backward(qkv)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        (qkv,) = ctx.saved_tensors
        B, sy, H, causal, p, seed, site = ctx.cfg
        d = qkv.shape[1] // 3
        d_out = d_out.contiguous().float()
        dqkv = torch.empty_like(qkv)
        e = qkv.element_size()
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        check(lib.egx_small_attention_bwd(base, 3 * d, base + d * e, 3 * d, base + 2 * d * e, 3 * d, ptr(d_out), d,
                                          gb, gb + d * e, gb + 2 * d * e, B, sy, sy, H, d // H, causal, p,
                                          C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        return dqkv, None, None, None, None, None, None, None


class CrossAttnSmallFn(torch.autograd.Function):
    """Attention of the target's queries (B * sy, d) onto the memory's packed (B * S, 2d) key/value rows -> (B * sy, d)."""

    @staticmethod
    def forward(ctx, q, kv, B: int, sy: int, S: int, H: int, p: float, seed: int, site: int):
        lib = _lib.load()
        q, kv = _dev_f32(q, "q"), _dev_f32(kv, "kv")
        d = q.shape[1]
        out = torch.empty((B * sy, d), dtype=torch.float32, device=q.device)
        e = kv.element_size()
        kb = kv.data_ptr()
        check(lib.egx_small_attention_fwd(ptr(q), d, kb, 2 * d, kb + d * e, 2 * d, ptr(out), d, B, sy, S, H, d // H, 0, float(p),
                                          C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        ctx.cfg = (B, sy, S, H, float(p), seed, site)
        ctx.save_for_backward(q, kv)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        q, kv = ctx.saved_tensors
        B, sy, S, H, p, seed, site = ctx.cfg
        d = q.shape[1]
        d_out = d_out.contiguous().float()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        e = kv.element_size()
        kb, gb = kv.data_ptr(), dkv.data_ptr()
        check(lib.egx_small_attention_bwd(ptr(q), d, kb, 2 * d, kb + d * e, 2 * d, ptr(d_out), d, ptr(dq), gb, gb + d * e,
                                          B, sy, S, H, d // H, 0, p, C.c_uint64(seed & (2**64 - 1)), site, _stream()))
        return dq, dkv, None, None, None, None, None, None, None
