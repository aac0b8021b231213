"""The EgoT2-g sequence decoder + vocabulary head on libegot2x.so: decode() of
HHI/models/multitask/task_prompt_model.py:260-269 and HOI/models/multitask/video_model_builder.py:150-159.

`embedding(y) * sqrt(d)` + positional encoding -> nn.TransformerDecoder of CustomDecoderLayer (post-LN: causal
self-attention over the 2..5 target tokens, cross-attention onto the encoder memory, ReLU FFN) -> `fc`. The nn modules
are parameter containers only; projections / FFN run through the MFMA GEMM (egx_linear_*), LayerNorms through
egx_layernorm_*, the two attentions through egx_small_attention_* and the embedding through egx_embed_pos_*."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F_egx

_SITE0 = 0x4000     # decoder dropout sites live above the encoder's (layer << 8 | site)


class DecoderMixin:
    def _egx_decode(self, y: torch.Tensor, encoded_x: torch.Tensor, *, embedding: nn.Embedding, pos_embed, decoder: nn.TransformerDecoder,
                    fc: nn.Linear, n_heads: int, p_drop: float) -> torch.Tensor:
        """y (B, sy) int64, encoded_x (S, B, d) decoder memory -> (sy, B, |V|) logits, as the reference's decode()."""
        S, B, d = encoded_x.shape
        sy = y.shape[1]
        if y.shape[0] != B:
            raise ValueError(f"target batch {y.shape[0]} != memory batch {B}")
        comp_model = getattr(self, "egx_compute", "f32")
        d_ff = decoder.layers[0].linear1.out_features
        post_ln = not any(getattr(layer, "norm_first", False) for layer in decoder.layers)
        if post_ln and F_egx.decoder_supported(comp_model, d, n_heads, d_ff, sy, S, len(decoder.layers)) and not getattr(self, "egx_composed_decoder", False):
            # ONE library call per direction (egx_decoder_fwd / egx_decoder_bwd): bf16 MFMA GEMMs over all B * sy target rows
            train = bool(self.training)
            meta = dict(n_layers=len(decoder.layers), n_heads=n_heads, d_ff=d_ff, ln_eps=decoder.layers[0].norm1.eps,
                        p_drop=p_drop if train else 0.0, p_pos=pos_embed.dropout.p if train else 0.0, training=train,
                        seed=self._egx_seed() if train else 0,
                        seed_ptr=(self._egx_seed_dev.data_ptr() if train and getattr(self, "_egx_seed_dev", None) is not None else 0))
            params = []
            for layer in decoder.layers:
                sa, ca = layer.self_attn, layer.multihead_attn
                params += [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.norm1.weight, layer.norm1.bias,
                           ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, layer.norm2.weight, layer.norm2.bias,
                           layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm3.weight, layer.norm3.bias]
            mem2d = encoded_x.permute(1, 0, 2).contiguous().view(B * S, d)
            out = F_egx.DecoderFn.apply(meta, y, mem2d, embedding.weight, pos_embed.pe[:sy, 0, :], *params, fc.weight, fc.bias)
            return out.view(B, sy, -1).permute(1, 0, 2)
        comp = "f32"        # (B * sy)-row GEMMs: negligible work, they always run the exact fp32 MFMA path
        comp_mem = getattr(self, "egx_compute", "f32")     # the K / V projection of the (B * S)-row memory follows the encoder's compute type
        train = bool(self.training)
        seed = self._egx_seed() if train else 0
        mem2d = encoded_x.permute(1, 0, 2).contiguous().view(B * S, d)          # batch-first rows b * S + s
        x = F_egx.EmbedPosFn.apply(y, embedding.weight, pos_embed.pe[:, 0, :], math.sqrt(d),
                                   pos_embed.dropout.p if train else 0.0, seed)
        for li, layer in enumerate(decoder.layers):
            if getattr(layer, "norm_first", False):
                raise ValueError("libegot2x implements the post-LN decoder layer only (norm_first=False)")
            site = lambda k: _SITE0 + (li << 8) + k  # noqa: E731
            p = p_drop if train else 0.0
            sa, ca = layer.self_attn, layer.multihead_attn
            qkv = F_egx.linear(x, sa.in_proj_weight, sa.in_proj_bias, comp)
            a = F_egx.SelfAttnSmallFn.apply(qkv, B, sy, n_heads, True, p, seed, site(1))
            a = F_egx.dropout(F_egx.linear(a, sa.out_proj.weight, sa.out_proj.bias, comp), p_drop, train, seed, site(2))
            x = F_egx.layer_norm_residual(x, a, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            q = F_egx.linear(x, ca.in_proj_weight[:d], ca.in_proj_bias[:d], comp)
            kv = F_egx.linear(mem2d, ca.in_proj_weight[d:], ca.in_proj_bias[d:], comp_mem)
            c = F_egx.CrossAttnSmallFn.apply(q, kv, B, sy, S, n_heads, p, seed, site(3))
            c = F_egx.dropout(F_egx.linear(c, ca.out_proj.weight, ca.out_proj.bias, comp), p_drop, train, seed, site(4))
            x = F_egx.layer_norm_residual(x, c, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
            h = F_egx.dropout(F_egx.linear(x, layer.linear1.weight, layer.linear1.bias, comp, relu=True), p_drop, train, seed, site(5))
            f = F_egx.dropout(F_egx.linear(h, layer.linear2.weight, layer.linear2.bias, comp), p_drop, train, seed, site(6))
            x = F_egx.layer_norm_residual(x, f, layer.norm3.weight, layer.norm3.bias, layer.norm3.eps)
        out = F_egx.linear(x, fc.weight, fc.bias, comp)                         # (B * sy, |V|)
        return out.view(B, sy, -1).permute(1, 0, 2)
