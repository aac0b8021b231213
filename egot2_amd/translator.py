"""Host-side core shared by every translator class: packs nn.Parameters into the C-ABI call of
libegot2x.so. The nn.TransformerEncoder / nn.LayerNorm / nn.Linear submodules are kept only as PARAMETER
CONTAINERS so that state_dict keys, shapes and default initialisation are byte-compatible with the reference
(SURVEY.md §8b); their forward() is never called on the product path.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import functional as F_egx
from .functional import EncoderSpec, SegmentSpec


class PositionalEncoding(nn.Module):
    """Sinusoidal table with the reference's buffer name and shape (max_len, 1, d)
    (HHI/models/ttm/model_taskspecific.py:131-151). The translator kernels read rows of `pe` directly; this
    module's forward is only used by the (torch) EgoT2-g sequence decoder."""

    def __init__(self, d_model, dropout=0.1, max_len=1000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        pe = pe.unsqueeze(0).transpose(0, 1)  # (max_len, 1, d_model)
        self.register_buffer('pe', pe)

    def forward(self, x):
        x = x + self.pe[:x.size(0), :]
        return self.dropout(x)


def encoder_layer_tensors(encoder: nn.TransformerEncoder) -> List[torch.Tensor]:
    """Flatten nn.TransformerEncoder parameters in the order of egx_layer's fields."""
    out = []
    for layer in encoder.layers:
        if getattr(layer, "norm_first", False):
            raise ValueError("libegot2x implements the post-LN encoder layer only (norm_first=False)")
        sa = layer.self_attn
        if sa.in_proj_weight is None:
            raise ValueError("separate q/k/v projection weights are not supported")
        out += [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]
    return out


class TranslatorMixin:
    """Adds the HIP encoder call to an nn.Module. `egx_compute` in {"f32", "bf16", "f32s"}; `egx_impl` in
    {"auto", "generic", "fused"}."""

    # default arithmetic (round 6): "f32s" = fp32 operands split exactly into three bf16 parts, six bf16 MFMAs per K-block, fp32 accumulation —
    # fp32-grade results (2e-6 on the logits against fp64, tests/test_gpu_translator.py::test_split_bf16_mode_is_fp32_grade) at 2.7x the matrix
    # rate of the exact fp32 MFMA; implemented by the per-clip d = 128 kernels, everywhere else it computes as "f32". set_compute("f32") is the
    # exact v_mfma_f32_16x16x4_f32 path, "bf16" the BASELINE.json configs[2..4] arithmetic.
    egx_compute: str = "f32s"
    egx_impl: str = "auto"
    egx_defer_small: bool = False      # staged backward for the all-reduce overlap (ddp.allreduce_gradients_overlapped)
    egx_deterministic: bool = False    # fixed-order reductions in the backward (egx_config.deterministic)
    _egx_step: int = 0

    def set_compute(self, compute: str = "f32", impl: str = "auto"):
        assert compute in F_egx.COMPUTE and impl in F_egx.IMPL
        self.egx_compute, self.egx_impl = compute, impl
        return self

    def set_deterministic(self, on: bool = True):
        """Bit-identical gradients run to run (same inputs, same seed): the fused per-clip backward replaces its fp32-atomic
        cross-workgroup sums by fixed-order slab reductions, the shape-generic backward routes split-K weight gradients and
        LayerNorm / bias / pooled-head parameter gradients through partial buffers with ordered sums; the wide bf16 path is
        deterministic by construction."""
        self.egx_deterministic = bool(on)
        return self

    def enable_weight_cache(self, frozen: bool = False):
        """Keep the MFMA-fragment-packed weight copies of the per-clip / tiled kernels in a persistent buffer and skip the packing launch
        of every forward whose weights did not change since the previous one (functional.WeightCache: decided on storage addresses,
        version counters and functional.note_weights_changed(); FusedAdam and GraphedStep call the latter). Writes that bypass those
        (`p.data.add_()`) need invalidate_weight_cache(). frozen=True additionally promises that the weights stay as they are while it is
        set (inference, a forward + backward benchmark): only then is the packing launch left out of a captured hipGraph as well, and the
        device-resident dropout seed is advanced by the backward (no launch in front of the forward at all)."""
        self._egx_wcache = F_egx.WeightCache(frozen=frozen)
        return self

    def disable_weight_cache(self):
        self._egx_wcache = None
        return self

    def invalidate_weight_cache(self):
        wc = getattr(self, "_egx_wcache", None)
        if wc is not None:
            wc.invalidate()
        return self

    def enable_device_seed(self, device=None):
        """Keep the dropout seed in device memory and advance it on the stream every training forward, so that a
        captured hipGraph (torch.cuda.graph around forward+backward) draws fresh masks on every replay."""
        dev = device or next(self.parameters()).device
        self._egx_seed_dev = torch.tensor([torch.initial_seed() & (2**62 - 1)], dtype=torch.int64, device=dev)
        return self

    def _egx_seed(self) -> int:
        # counter-based: one fresh dropout key per forward, no device sync
        self._egx_step += 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._egx_step * 0xD1B54A32D192ED03) & (2**63 - 1)

    def _egx_encode(self, feats: Sequence[torch.Tensor], segments: List[SegmentSpec], *, encoder: nn.TransformerEncoder,
                    ln: nn.LayerNorm, projs: Sequence[Optional[nn.Linear]], task_embed: Optional[torch.Tensor],
                    pos_table: Optional[torch.Tensor], p_drop: float, p_pos: float = 0.0, p_feat: float = 0.0,
                    head=None, out_tokens: int = 0, ce=None, token_ce=None) -> torch.Tensor:
        """head = (nn.LayerNorm, nn.Linear): evaluate the pooled head with the encoder and return logits (B, n_out).
        out_tokens = T > 0: return only the first T tokens of every clip, (B, T, d) (in-kernel on the fused path).
        ce = (target, class_weight | None) with a head: also evaluate nn.CrossEntropyLoss(weight)(logits, target) inside the forward
        (egx_ce) and return (logits, loss).
        token_ce = (fc_weight, fc_bias | None, target, class_weight | None) with out_tokens: return (loss, logits, probs, pred, correct) of the
        per-token classifier + weighted cross entropy on the returned tokens instead of the tokens (functional.encoder_token_ce)."""
        layer0 = encoder.layers[0]
        d = ln.normalized_shape[0]
        seed_dev = getattr(self, "_egx_seed_dev", None)
        wcache = getattr(self, "_egx_wcache", None)
        # device seed: advanced by the forward's first launch (1) or, with a frozen weight cache (no launch in front of the forward), by the
        # backward behind its last reader (2)
        adv = 0 if (seed_dev is None or not self.training) else (2 if (wcache is not None and wcache.frozen and torch.is_grad_enabled()) else 1)
        impl = self.egx_impl    # "auto": functional.EncoderFn steers around the fused kernels when a learned `pe` needs a gradient
        spec = EncoderSpec(d_model=d, n_heads=layer0.self_attn.num_heads, d_ff=layer0.linear1.out_features,
                           n_layers=len(encoder.layers), segments=segments, ln_eps=ln.eps,
                           compute=self.egx_compute, impl=impl,
                           p_drop=p_drop, p_pos=p_pos, p_feat=p_feat,
                           training=bool(self.training), seed=self._egx_seed() if self.training else 0,
                           seed_ptr=seed_dev.data_ptr() if seed_dev is not None else 0,
                           head_n_out=head[1].out_features if head is not None else 0,
                           advance_seed=adv,   # fresh masks per (replayed) step
                           defer_small=bool(self.egx_defer_small), deterministic=bool(self.egx_deterministic),
                           out_tokens=int(out_tokens), wcache=wcache, ce=ce is not None)
        proj_t = []
        for s, p in zip(segments, projs):
            if s.has_proj:
                proj_t += [p.weight, p.bias]
        head_t = (head[0].weight, head[0].bias, head[1].weight, head[1].bias) if head is not None else ()
        if ce is not None and head is None:
            raise ValueError("the fused cross entropy needs the pooled head")
        if token_ce is not None:
            if head is not None or not out_tokens:
                raise ValueError("the token classifier works on the first out_tokens tokens of every clip, without the pooled head")
            return F_egx.encoder_token_ce(spec, list(feats), task_embed, pos_table, ln.weight, ln.bias, proj_t,
                                          encoder_layer_tensors(encoder), *token_ce)
        return F_egx.encoder(spec, list(feats), task_embed, pos_table, ln.weight, ln.bias, proj_t,
                             encoder_layer_tensors(encoder), head_t, ce=ce)
